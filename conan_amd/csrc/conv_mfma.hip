// conv_mfma: channel-last causal / shifted 1-D convolution as an implicit GEMM on the gfx950
// f32 MFMA (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD).
//
// GEMM view:  M = n*T output rows (slot, time), N = Cout, K = ktaps*Cin.
//   A[m][(j,ci)] = f(x[slot][t + j*dil - pad_left][ci])   gathered per tap from the activation rings
//   B[(j,ci)][co] = packed weights [tap][ci/4][co][4]
// Block = 512 threads = 8 waves, two per SIMD, specialised: waves 0-3 issue nothing but fragment reads and
// MFMAs; waves 4-7 stage the tiles.  Per K-step (one tap, KS input channels) the loader waves move an A tile
// [TM][KS] (gathered ring rows) and a W tile [KS][TN] global -> LDS directly (global_load_lds_dwordx4: no VGPR
// round trip, no ds_write).  A wave instruction lands lane-linear in LDS, so the A tile is unpadded and bank
// conflicts are removed by an XOR swizzle of its 16-byte chunks, applied to the per-lane SOURCE address and again
// at the fragment read.  Three LDS buffers: loads run two K-steps ahead, retired by counted s_waitcnt vmcnt(N) and
// one raw s_barrier per step shared by both roles.  The input is never transformed on the way (the DMA cannot):
// producers store activated tensors; a second K-loop variant applies LeakyReLU on the A fragments when asked.
// Fragments are read with ds_read_b128: a K-chunk of 8 feeds 4 MFMAs, lanes 0-31 carrying k 0..3 and
// lanes 32-63 k 4..7 (the MFMA's two k-slots), so one 16-byte LDS read per operand serves 4 matrix
// instructions; the two fragment register sets are double-buffered by hand.
// Output tile D[time][co] keeps co on the lane; each 32x32 accumulator is transposed through a wave-private LDS
// patch so that stores are 16-byte channel-last accesses; the pixel shuffle of CausalUpsampleBlock3 is a pure
// address remap of that store (weights pre-permuted).
// KS = 32 for the large streaming tiles; KS = 64/128 for the small-M (latency-bound) tiles, where a longer K-step
// amortises the per-step load latency and barrier.  Blocks are persistent over a host-made balanced tile list.
#include <type_traits>

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <queue>
#include <vector>

#include "kernels.h"

namespace cnk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case ACT_LRELU: return v > 0.f ? v : v * slope;
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    case ACT_TANH: return tanhf(v);
    default: return v;
  }
}

__device__ __forceinline__ unsigned tref_row(const TRef& r, int slot, const int* pos, int t) {
  if (r.mode == 0) {
    unsigned p = pos ? (unsigned)pos[slot] : 0u;
    return (p * (unsigned)r.rate + (unsigned)(r.off + t)) & (unsigned)r.lmask;
  }
  return (unsigned)(r.off + t);
}

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

template <int TM, int TN, int WK, int KS>
struct ConvLds {
  static constexpr int NBUF = 3;
  static constexpr int LDA = KS;
  static constexpr int STAGE = NBUF * (TM * LDA + KS * TN);
  static constexpr int EPI_LD = 36;                 // epilogue transpose patch: 32 rows x 36 floats per compute wave
  // (only the compute waves with wk == 0 run the epilogue - waves 0 .. 4 / WK - 1: the K-split builds need a patch per such wave, not
  // per compute wave; the 32 x 32 x K4 build went from 126 to 112.5 KB, which leaves room for a single-tile decoder workgroup - 36.5 KB -
  // on the same CU: small-batch pipelined steps)
  static constexpr int PATCH = (4 / WK) * 32 * EPI_LD;
  static constexpr int RED = (WK > 1) ? (WK - 1) * TM * TN : 0;   // split-K partial tiles of waves wk > 0
  static constexpr int FLAG = STAGE + PATCH + RED;        // the split-K "reducer" flag
  static constexpr int TOTAL = STAGE + PATCH + RED + 4;
};

// One output tile (rows m0.., columns n0..) of one problem.  `gbuf` is the LDS ring position of the tile's first
// K-step: it is carried from tile to tile so that a persistent block's loader waves can start the next tile's
// loads while the matrix waves are still in the previous tile's epilogue.
template <int TM, int TN, int WM, int WN, int WK, int KS>
__device__ __forceinline__ void conv_tile(const ConvArgs& a, const int m0, const int n0, float* lds, int& gbuf, const int tile,
                                          const int kslice, const int nslices, float* slab, int* counter, const int fenced) {
  static_assert(WM * WN * WK == 4, "4 compute waves per block");
  static_assert(KS == 32 || KS == 64 || KS == 128, "K-step");
  constexpr int RM = TM / WM / 32;
  constexpr int RN = TN / WN / 32;
  constexpr int SB = KS / 32;                 // 32-channel sub-blocks per K-step
  constexpr int WV = (KS / 4 * TN) / 256;     // W float4 staged per loader thread
  constexpr int NKQ = KS / 8;                 // 8-deep K chunks per step
  // Tiles go global -> LDS directly (global_load_lds, no VGPR round trip, no ds_write); the LDS image
  // of a wave instruction is lane-linear, so A rows are unpadded and bank conflicts are removed by an XOR swizzle
  // of the 16-byte chunks applied on the per-lane SOURCE address and again on the fragment read; three buffers
  // (loads run two K-steps ahead).
  constexpr int NBUF = 3;
  constexpr int LDA = KS;
  constexpr int CPR = KS / 4;                  // 16-byte chunks per A row
  constexpr int RPI = 64 / CPR;                // A rows covered by one wave-wide 1 KiB load
  constexpr int NA = TM / RPI / 4;             // A load instructions per loader wave per K-step
  constexpr int A_FLOATS = TM * LDA;
  constexpr int W_FLOATS = KS * TN;
  using L = ConvLds<TM, TN, WK, KS>;
  static_assert(L::RED == ((WK > 1) ? (WK - 1) * WM * WN * RM * RN * 16 * 64 : 0), "split-K reduction region");
  constexpr int STAGE_FLOATS_TOTAL = L::STAGE;      // [staging ring | epilogue patches | split-K reduction]
  constexpr int EPI_LD = L::EPI_LD;
  float* const red = lds + L::STAGE + L::PATCH;

  // ---- launch-uniform scalars, read once (keeps the K loop free of kernarg re-loads)
  const int T = a.T, nslot = a.n, ktaps = a.ktaps, dil = a.dil, Cin = a.Cin, Cout = a.Cout;
  const int Mtot = nslot * T;
  const int* __restrict__ slots = a.slots;
  const int* __restrict__ posp = a.pos;
  const float* __restrict__ wbase = a.w;

  // Wave specialisation: waves 0-3 (one per SIMD) only read fragments and issue MFMAs; waves 4-7 (their SIMD
  // partners) only stage tiles global -> registers -> LDS.  The two instruction streams interleave in hardware,
  // so address arithmetic, loads, the input transform and the LDS writes run in the shadow of the (dependent,
  // 64-cycle) MFMA chain instead of in front of it.
  const int tid = threadIdx.x;
  const bool is_loader = tid >= 256;
  const int ltid = tid & 255;
  const int lane = tid & 63;
  const int wave = (tid >> 6) & 3;
  const int wn = wave % WN;
  const int wm = (wave / WN) % WM;
  const int wk = (WK == 1) ? 0 : wave / (WN * WM);
  const int l31 = lane & 31;
  const int lh = lane >> 5;

  // ---- tile row -> (batch index i, time t).  A tile that fits inside one stream's T rows touches at most two
  // streams: their slot / position are fetched once; short-T layers (frame-rate tensors) divide per row.
  const int i0 = m0 / T;
  const int t0 = m0 - i0 * T;
  const bool fast = T >= TM;
  const int i1 = (i0 + 1 < nslot) ? i0 + 1 : i0;
  const int slotA = slots ? slots[i0] : i0, slotB = slots ? slots[i1] : i1;
  const int posA = posp ? posp[slotA] : 0, posB = posp ? posp[slotB] : 0;
  auto rowmap = [&](int ml, int& i, int& t, int& slot, int& pv) __attribute__((always_inline)) {
    if (fast) {
      const int tt = t0 + ml;
      const bool w = tt >= T;
      i = w ? i1 : i0; t = w ? tt - T : tt; slot = w ? slotB : slotA; pv = w ? posB : posA;
    } else {
      const int m = m0 + ml;
      i = m / T; t = m - i * T;
      i = i < nslot ? i : nslot - 1;
      slot = slots ? slots[i] : i; pv = posp ? posp[slot] : 0;
    }
  };

  const int ncb32 = a.Cin_pad >> 5;                 // 32-channel blocks per tap
  const int ncb = (ncb32 + SB - 1) / SB;            // K-steps per tap
  const int nks_all = ktaps * ncb;
  // inter-block split-K is compiled into the small-M shapes (every tile of a launch) and into the 64-column K-step-32
  // streaming shapes (the tail tiles of a launch); conv_cfg_splitk() tells the host which
  constexpr bool SK = (TM == 32) || (TN == 64 && KS == 32 && WK == 1);
  // inter-block split-K: this block owns K-steps [ks_lo, ks_lo + nks) of the tile; the partial tiles are combined by the
  // last-arriving block (ticket counter), in slice order, so the sum is reproducible
  const int ks_lo = SK ? (int)((long long)nks_all * kslice / nslices) : 0;
  const int nks = SK ? (int)((long long)nks_all * (kslice + 1) / nslices) - ks_lo : nks_all;
  int* const sflag = reinterpret_cast<int*>(lds + L::FLAG);
  auto block_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier();
  };
  // every wave of the block takes part in the two barriers of the split-K hand-off (X1: partial tiles written,
  // X2: "this block is the reducer" flag published through LDS)
  auto splitk_idle = [&]() __attribute__((always_inline)) {
    if constexpr (SK) { if (nslices > 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); block_barrier(); block_barrier(); } }
  };

  if (is_loader) {
    // ================================================================= loader waves
    {
#ifndef CK_NO_LOADER_PRIO
      __builtin_amdgcn_s_setprio(3);       // few instructions, all on the critical path of the next barrier
#endif
      const int lwave = __builtin_amdgcn_readfirstlane(ltid >> 6);
      const int prow = lane / CPR;                     // row inside one wave instruction
      const int pchunk = lane % CPR;                   // physical chunk written by this lane
      // Per-lane address state is 32-bit float offsets from the (uniform) tensor base, so a load is "scalar base +
      // VGPR offset" and a K-step costs ~5 vector instructions per A load: the loader waves share their SIMD's issue
      // slots with MFMA-saturated matrix waves, and every VALU instruction here shows up in the step time.
      int abrow[NA], aoff[NA], amask[NA], amask_last[NA];     // masks: all ones for lanes that load, 0 otherwise
      const bool xring = a.x.mode == 0;
      const float* const xb = a.x.base;
      {
        const int xrate = a.x.rate, xoff = a.x.off - a.pad_left;
        const long long xss = a.x.slot_stride;
#pragma unroll
        for (int u = 0; u < NA; ++u) {
          const int ml = (u * 4 + lwave) * RPI + prow;
          int i, t, slot, pv;
          rowmap(ml, i, t, slot, pv);
          abrow[u] = (xring ? pv * xrate : 0) + xoff + t;
          const int sw = (KS == 32) ? ((ml >> 1) & 7) : (ml & 15);
          const int acol = (pchunk ^ sw) * 4;          // logical channel offset inside the K-step
          aoff[u] = ((int)((long long)(xring ? slot : i) * xss) + acol) * 4;      // bytes
          // rows past the tile and (when Cin is not a multiple of the K-step) channels past Cin read offset 0: their
          // products are discarded (row never stored) or multiplied by the zero-padded weight rows
          amask[u] = (m0 + ml) < Mtot ? -1 : 0;
          amask_last[u] = ((Cin % KS) != 0 && acol >= (Cin % KS)) ? 0 : amask[u];
        }
      }
      const int xC4 = a.x.C * 4;
      const int xmask = xring ? a.x.lmask : -1;
      const int ci4n = a.Cin_alloc >> 2;
      const int cb_last = (Cin % KS) != 0 ? Cin / KS : 0x7fffffff;
      // weights are packed per group of 64 output columns: [n/64][tap][Cin_alloc/4][64][4] (1 KiB rows)
      const float* wgrp = wbase + ((long long)(n0 >> 6) * ktaps * ci4n * 64 + (n0 & 63)) * 4;
      int woff[WV];
#pragma unroll
      for (int v = 0; v < WV; ++v) { const int idx = ltid + 256 * v; const int kq4 = idx / TN, co = idx - kq4 * TN; woff[v] = (kq4 * 64 + co) * 4; }
      int jn = ks_lo % ktaps, cbn = ks_lo / ktaps;
      auto issue = [&](int buf) __attribute__((always_inline)) {
        const int j = jn, cb = cbn;
        float* As = lds + buf * (A_FLOATS + W_FLOATS);
        float* Ws = As + A_FLOATS;
        const int jd = j * dil, cbk4 = cb * KS * 4;
        const bool lastcb = cb >= cb_last;
#pragma unroll
        for (int u = 0; u < NA; ++u) {
          const int r = (abrow[u] + jd) & xmask;
          const unsigned off = (unsigned)((__mul24(r, xC4) + aoff[u] + cbk4) & (lastcb ? amask_last[u] : amask[u]));
          __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(reinterpret_cast<const char*>(xb) + off), As + ((u * 4 + lwave) * RPI) * LDA, 16, 0, 0);
        }
        const float* wstep = wgrp + (long long)(j * ci4n + cb * (KS / 4)) * 256;
#pragma unroll
        for (int v = 0; v < WV; ++v) {
          const int o = woff[v];
          __builtin_amdgcn_global_load_lds(wstep + o, Ws + (v * 4 + lwave) * 256, 16, 0, 0);
        }
        if (++jn == ktaps) { jn = 0; ++cbn; }
      };
      constexpr int NPER = NA + WV;          // loads per wave per K-step
      static_assert(NPER <= 31, "vmcnt budget");
#ifdef CK_STAMPS
      unsigned long long l_vm = 0, l_bar = 0, l_iss = 0, lt = __builtin_amdgcn_s_memtime();
#define CK_LSTAMP(acc) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - lt; lt = t_; }
#else
#define CK_LSTAMP(acc)
#endif
      const int g0 = gbuf;
      gbuf = (gbuf + nks) % 3;
      issue(g0);
      if (nks > 1) issue((g0 + 1) % 3);
      CK_LSTAMP(l_iss)
      for (int ks = 0; ks < nks; ++ks) {
        // step ks must have landed; the loads of step ks+1 may stay in flight
        if (ks + 1 < nks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CK_LSTAMP(l_vm)
        __builtin_amdgcn_s_barrier();        // B(ks): tile ks visible; every matrix wave is done with tile ks-1
        CK_LSTAMP(l_bar)
        if (ks + 2 < nks) issue((g0 + ks + 2) % 3);   // overwrites the buffer of K-step ks-1
        CK_LSTAMP(l_iss)
      }
#ifdef CK_STAMPS
      if (a.dbg && tile == 1 && ltid == 0) { a.dbg[8] = l_vm; a.dbg[9] = l_bar; a.dbg[10] = l_iss; }
#endif
#undef CK_LSTAMP
      if constexpr (WK > 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }
      splitk_idle();
      return;
    }
  }

  // ===================================================================== compute waves
  f32x16 acc[RM][RN];
#pragma unroll
  for (int rm = 0; rm < RM; ++rm)
#pragma unroll
    for (int rn = 0; rn < RN; ++rn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[rm][rn][e] = 0.f;

  // Small-M shapes (one 32 x 32 tile per wave group: latency-bound launches of 10 us): the operands are requested in FRONT of the K
  // loop by the wave that will run the epilogue - the matrix waves issue no vector loads in the loop, so nothing waits for them -
  // instead of behind the split-K hand-off, where each `if (has_res)` / `if (bias)` block was a round trip of its own.
  // (only the 32 x 32 shape: the 32 x 64 shape is ups.0's at serving sizes, where its workgroups share CUs with the decoder's persistent
  // launch - 2 x 139 + 128 registers per SIMD lane fit, 2 x 231 did not, and the pipelined step lost 4 %)
  constexpr bool EPF = (TM == 32 && TN == 32);
  int pf_ri[4], pf_rt[4], pf_rslot[4], pf_rpos[4], pf_co4 = 0; unsigned pf_ok = 0; float pf_mk[4]; float4 pf_rv[4], pf_bq;
  if constexpr (EPF) {
    static_assert(!EPF || (RM == 1 && RN == 1), "one tile per wave");
    if (wk == 0) {
#include "conv_mfma_epi.inc"
  // the epilogue's per-row and per-column operands (row -> stream / time / slot / frame counter, masks; residual, style vector, bias)
  auto row_ops = [&](const int rm, int (&ri)[4], int (&rt)[4], int (&rslot)[4], int (&rpos)[4], unsigned& okbits, float (&mkv)[4]) __attribute__((always_inline)) {
    const int er = lane >> 3;
      // (1) the lane's 4 rows: tile row er + 8g -> (batch index, time, slot, position)
      okbits = 0;
      if (fast) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int ml = (wm * RM + rm) * 32 + er + 8 * gq;
          const int tt = t0 + ml;
          const bool w = tt >= T;
          ri[gq] = w ? i1 : i0; rt[gq] = w ? tt - T : tt; rslot[gq] = w ? slotB : slotA; rpos[gq] = w ? posB : posA;
          okbits |= ((m0 + ml) < Mtot ? 1u : 0u) << gq;
        }
      } else {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int m = m0 + (wm * RM + rm) * 32 + er + 8 * gq;
          const int i = m / T;
          rt[gq] = m - i * T;
          okbits |= (m < Mtot ? 1u : 0u) << gq;
          ri[gq] = i < nslot ? i : nslot - 1;
        }
        if (slots) {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) rslot[gq] = slots[ri[gq]];
        } else {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) rslot[gq] = ri[gq];
        }
        if (posp) {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) rpos[gq] = posp[rslot[gq]];
        } else {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) rpos[gq] = 0;
        }
      }
      if (lens) {
        int rl[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) rl[gq] = lens[ri[gq]];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) if (rt[gq] >= rl[gq]) okbits &= ~(1u << gq);
      }
      mkv[0] = mkv[1] = mkv[2] = mkv[3] = 1.f;
      if (a.has_m1) {
        const TRef q = a.m1;
        const bool qring = q.mode == 0;
        const int qmask = qring ? q.lmask : -1;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int sidx = qring ? rslot[gq] : ri[gq];
          mkv[gq] = q.base[(long long)sidx * q.slot_stride + (((qring ? rpos[gq] * q.rate : 0) + q.off + rt[gq]) & qmask)];
        }
      }
      if (a.has_m2) {
        const TRef q = a.m2;
        const bool qring = q.mode == 0;
        const int qmask = qring ? q.lmask : -1;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int sidx = qring ? rslot[gq] : ri[gq];
          mkv[gq] *= q.base[(long long)sidx * q.slot_stride + (((qring ? rpos[gq] * q.rate : 0) + q.off + rt[gq]) & qmask)];
        }
      }
  };
  auto col_ops = [&](const int rn, const int (&ri)[4], const int (&rt)[4], const int (&rslot)[4], const int (&rpos)[4], int& co4, int& cc, bool& cok,
                     float4 (&rv)[4], float4& bq) __attribute__((always_inline)) {
    const int ec4 = lane & 7;
        co4 = n0 + (wn * RN + rn) * 32 + ec4 * 4;      // first of the lane's 4 channels
        cok = co4 < Cout;
        cc = cok ? co4 : 0;
        // (2) residual + style vector (16-byte loads where the layout allows)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) rv[gq] = f4zero();
        if (has_res) {
          const TRef rr = a.res;
          const bool rring = rr.mode == 0;
          const int rmask = rring ? rr.lmask : -1;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int sidx = rring ? rslot[gq] : ri[gq];
            const int row = ((rring ? rpos[gq] * rr.rate : 0) + rr.off + rt[gq]) & rmask;
            const float* p = rr.base + (long long)sidx * rr.slot_stride + row * rr.C + cc;
            if (vec_ok && (rr.C & 3) == 0) rv[gq] = *reinterpret_cast<const float4*>(p);
            else { rv[gq].x = p[0]; rv[gq].y = cc + 1 < Cout ? p[1] : 0.f; rv[gq].z = cc + 2 < Cout ? p[2] : 0.f; rv[gq].w = cc + 3 < Cout ? p[3] : 0.f; }
          }
        }
        if (has_bvec) {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const float* bv = a.bvec + (long long)rslot[gq] * a.bvec_stride + cc;
            rv[gq].x += bv[0]; rv[gq].y += cc + 1 < Cout ? bv[1] : 0.f; rv[gq].z += cc + 2 < Cout ? bv[2] : 0.f; rv[gq].w += cc + 3 < Cout ? bv[3] : 0.f;
          }
        }
        bq = f4zero();
        if (a.bias) { bq.x = a.bias[cc]; bq.y = a.bias[cc + 1]; bq.z = a.bias[cc + 2]; bq.w = a.bias[cc + 3]; }   // bias is padded to Cout_pad
  };
      row_ops(0, pf_ri, pf_rt, pf_rslot, pf_rpos, pf_ok, pf_mk);
      int cc_; bool cok_;
      col_ops(0, pf_ri, pf_rt, pf_rslot, pf_rpos, pf_co4, cc_, cok_, pf_rv, pf_bq);
    }
  }
  const float neg_mul_c = a.in_act == ACT_LRELU ? a.in_slope : 1.0f;
  int bufc = gbuf;
  gbuf = (gbuf + nks) % NBUF;
#ifdef CK_STAMPS
  unsigned long long c_wait = 0, c_comp = 0, ct = __builtin_amdgcn_s_memtime();
  const unsigned long long c_begin = ct;
  const unsigned long long c_rt0 = __builtin_amdgcn_s_memrealtime();
#define CK_CSTAMP(acc) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - ct; ct = t_; }
#else
#define CK_CSTAMP(acc)
#endif
  // The direct-to-LDS loader cannot transform on the way, so a LeakyReLU on the conv input has to be applied to the A
  // fragments here - a dozen VALU ops per k-group that cost a quarter of the matrix rate (tools/kloop_rate).  The
  // loop is therefore compiled twice and the hot layers are fed pre-activated tensors (ConvArgs::y2_base).
  auto kloop = [&](auto xf_tag) __attribute__((always_inline)) {
    constexpr bool XF = decltype(xf_tag)::value;
    for (int ks = 0; ks < nks; ++ks) {
      const float* As = lds + bufc * (A_FLOATS + W_FLOATS);
      const float* Ws = As + A_FLOATS;
      if (++bufc == NBUF) bufc = 0;
      asm volatile("s_barrier" ::: "memory");   // B(ks): tile ks landed in LDS (clobber: no LDS read may move above it)
      CK_CSTAMP(c_wait)
      // Fragments are double-buffered by hand: the ds_reads of k-group kc+1 are issued before the MFMAs of k-group kc,
      // so only the first read after the barrier is exposed (left to itself the compiler reuses one register set and
      // every k-group starts with a full LDS round trip - a third of the step time).
      constexpr int NKC = NKQ / WK;
      float4 af[2][RM], bf[2][RN];
      auto load_frag = [&](const int kc, float4 (&fa)[RM], float4 (&fb)[RN]) __attribute__((always_inline)) {
        const int kq = kc * WK + wk;      // fixed trip count: no divergent control flow around the MFMAs
  #pragma unroll
        for (int rm = 0; rm < RM; ++rm) {
          const int R = (wm * RM + rm) * 32 + l31;
          const int sw = (KS == 32) ? ((R >> 1) & 7) : (R & 15);
          fa[rm] = *reinterpret_cast<const float4*>(As + R * LDA + (((kq * 2 + lh) ^ sw) * 4));
        }
  #pragma unroll
        for (int rn = 0; rn < RN; ++rn)
          fb[rn] = *reinterpret_cast<const float4*>(Ws + ((kq * 2 + lh) * TN + (wn * RN + rn) * 32 + l31) * 4);
      };
      load_frag(0, af[0], bf[0]);
  #pragma unroll
      for (int kc = 0; kc < NKC; ++kc) {
        if (kc + 1 < NKC) load_frag(kc + 1, af[(kc + 1) & 1], bf[(kc + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);   // keep the reads above this k-group's MFMAs
        float4 (&fa)[RM] = af[kc & 1];
        float4 (&fb)[RN] = bf[kc & 1];
        if constexpr (XF) {
          // the loader cannot transform on the way: LeakyReLU of the conv input is applied here
  #pragma unroll
          for (int rm = 0; rm < RM; ++rm) {
            float4 v = fa[rm];
            v.x *= v.x > 0.f ? 1.0f : neg_mul_c; v.y *= v.y > 0.f ? 1.0f : neg_mul_c;
            v.z *= v.z > 0.f ? 1.0f : neg_mul_c; v.w *= v.w > 0.f ? 1.0f : neg_mul_c;
            fa[rm] = v;
          }
        }
  #pragma unroll
        for (int rm = 0; rm < RM; ++rm)
  #pragma unroll
          for (int rn = 0; rn < RN; ++rn) {
            acc[rm][rn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[rm].x, fb[rn].x, acc[rm][rn], 0, 0, 0);
            acc[rm][rn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[rm].y, fb[rn].y, acc[rm][rn], 0, 0, 0);
            acc[rm][rn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[rm].z, fb[rn].z, acc[rm][rn], 0, 0, 0);
            acc[rm][rn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[rm].w, fb[rn].w, acc[rm][rn], 0, 0, 0);
          }
      }
  #ifdef CK_STAMPS
      asm volatile("s_nop 0" ::"v"(acc[0][0][0]));
  #endif
      CK_CSTAMP(c_comp)
    }
  };
  if (a.in_act == ACT_LRELU) kloop(std::true_type{}); else kloop(std::false_type{});
#ifdef CK_STAMPS
  const unsigned long long c_kend = __builtin_amdgcn_s_memtime();
#endif

  // ---- intra-block split-K reduction through LDS
  if constexpr (WK > 1) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier();
    constexpr int PER_WAVE = RM * RN * 16 * 64;
    if (wk > 0) {
      float* dst = red + ((wk - 1) * WM * WN + wm * WN + wn) * PER_WAVE;
#pragma unroll
      for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int rn = 0; rn < RN; ++rn)
#pragma unroll
          for (int e = 0; e < 16; ++e) dst[((rm * RN + rn) * 16 + e) * 64 + lane] = acc[rm][rn][e];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier();
    if (wk > 0) { splitk_idle(); return; }
#pragma unroll
    for (int k2 = 1; k2 < WK; ++k2) {
      const float* src = red + ((k2 - 1) * WM * WN + wm * WN + wn) * PER_WAVE;
#pragma unroll
      for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int rn = 0; rn < RN; ++rn)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[rm][rn][e] += src[((rm * RN + rn) * 16 + e) * 64 + lane];
    }
  }

  // ---- inter-block split-K hand-off.  MI355X: per-XCD L2s are not coherent; the partial tiles travel as agent-scope
  // write-through stores / sc1 loads (relaxed atomics), ordered by s_waitcnt + the ticket - no release / acquire fence,
  // i.e. no write-back and invalidate of the whole L2 per tile (same scheme as the fused Emformer's cluster exchange).
#ifdef CK_STAMPS
  unsigned long long* const rdbg = (a.dbg && blockIdx.x < 512) ? a.dbg + 1100 + blockIdx.x * 8 : nullptr;
#define CK_RSTAMP(i) do { if (rdbg && tid == 0) rdbg[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CK_RSTAMP(i) do { } while (0)
#endif
  if (SK && nslices > 1) {
    CK_RSTAMP(0);
    constexpr int PER_WAVE = RM * RN * 16 * 64;
    constexpr int PER_TILE = WM * WN * PER_WAVE;
    float* mine = slab + (long long)kslice * PER_TILE + (wm * WN + wn) * PER_WAVE;
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
      for (int rn = 0; rn < RN; ++rn)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          __hip_atomic_store(mine + ((rm * RN + rn) * 16 + e) * 64 + lane, acc[rm][rn][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its (write-through) stores
    if (fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    CK_RSTAMP(1);
    block_barrier();                                       // X1
    if (tid == 0) {
      const int ticket = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == nslices - 1;
      if (last) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ready for the next launch
      *sflag = last;
    }
    block_barrier();                                       // X2
    CK_RSTAMP(2);
    if (*sflag == 0) return;
    if (fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // reducer: sum the partial tiles in slice order (own slice from memory too: identical bits, fixed order)
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
      for (int rn = 0; rn < RN; ++rn)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[rm][rn][e] = 0.f;
    // (QB slices per round trip - 4 for the 32 x 32 shape: 166 registers; 8 slices at a time measured the same and took 231 -: left as one slice per loop iteration the loads of slice q + 1 are issued behind the adds of slice q - one
    // round trip to memory per slice, 1-2 us each, on the launch's critical path.  Slices past the last re-read the last one and are dropped.)
    constexpr int QB = (TM == 32 && TN == 32) ? 4 : 2;      // (the streaming shapes hold 128 / 160 registers: two slices at a time)
    for (int q0 = 0; q0 < nslices; q0 += QB) {
      float pv[QB][RM * RN * 16];
#pragma unroll
      for (int j = 0; j < QB; ++j) {
        const int q = q0 + j < nslices ? q0 + j : nslices - 1;
        const float* src = slab + (long long)q * PER_TILE + (wm * WN + wn) * PER_WAVE;
#pragma unroll
        for (int f = 0; f < RM * RN * 16; ++f) pv[j][f] = __hip_atomic_load(src + f * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int j = 0; j < QB; ++j) {
        const bool in = q0 + j < nslices;
#pragma unroll
        for (int rm = 0; rm < RM; ++rm)
#pragma unroll
          for (int rn = 0; rn < RN; ++rn)
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float t = acc[rm][rn][e] + pv[j][(rm * RN + rn) * 16 + e]; acc[rm][rn][e] = in ? t : acc[rm][rn][e]; }
      }
    }
    CK_RSTAMP(3);
  }

  // ---- epilogue.  Each 32x32 accumulator tile (column on the lane, 16 rows in registers) is transposed through a
  // wave-private LDS patch so that every lane owns 4 consecutive output channels of one row: residual loads and
  // output stores become 16-byte accesses (4 store instructions per tile instead of 16 -- the store tail is
  // issue-bound, not bandwidth-bound) and the row -> stream/time/slot arithmetic is done for 4 rows per lane
  // instead of 16.  Every run-time option and the activation are uniform branches around straight-line passes, so
  // the loads of a pass are issued back-to-back and waited for once.
#include "conv_mfma_epi.inc"
  auto epilogue = [&](auto act_tag) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_tag)::value;
    float* const patch = lds + STAGE_FLOATS_TOTAL + wave * (32 * EPI_LD);
    const int er = lane >> 3, ec4 = lane & 7;
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) {
      int ri[4], rt[4], rslot[4], rpos[4];
      unsigned okbits = 0;
      float mkv[4] = {1.f, 1.f, 1.f, 1.f};
      if constexpr (EPF) {      // (requested in front of the K loop: row_ops / col_ops above)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) { ri[gq] = pf_ri[gq]; rt[gq] = pf_rt[gq]; rslot[gq] = pf_rslot[gq]; rpos[gq] = pf_rpos[gq]; mkv[gq] = pf_mk[gq]; }
        okbits = pf_ok;
      } else {
        // (1) the lane's 4 rows: tile row er + 8g -> (batch index, time, slot, position)
        if (fast) {
  #pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int ml = (wm * RM + rm) * 32 + er + 8 * gq;
            const int tt = t0 + ml;
            const bool w = tt >= T;
            ri[gq] = w ? i1 : i0; rt[gq] = w ? tt - T : tt; rslot[gq] = w ? slotB : slotA; rpos[gq] = w ? posB : posA;
            okbits |= ((m0 + ml) < Mtot ? 1u : 0u) << gq;
          }
        } else {
  #pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int m = m0 + (wm * RM + rm) * 32 + er + 8 * gq;
            const int i = m / T;
            rt[gq] = m - i * T;
            okbits |= (m < Mtot ? 1u : 0u) << gq;
            ri[gq] = i < nslot ? i : nslot - 1;
          }
          if (slots) {
  #pragma unroll
            for (int gq = 0; gq < 4; ++gq) rslot[gq] = slots[ri[gq]];
          } else {
  #pragma unroll
            for (int gq = 0; gq < 4; ++gq) rslot[gq] = ri[gq];
          }
          if (posp) {
  #pragma unroll
            for (int gq = 0; gq < 4; ++gq) rpos[gq] = posp[rslot[gq]];
          } else {
  #pragma unroll
            for (int gq = 0; gq < 4; ++gq) rpos[gq] = 0;
          }
        }
        if (lens) {
          int rl[4];
  #pragma unroll
          for (int gq = 0; gq < 4; ++gq) rl[gq] = lens[ri[gq]];
  #pragma unroll
          for (int gq = 0; gq < 4; ++gq) if (rt[gq] >= rl[gq]) okbits &= ~(1u << gq);
        }
        if (a.has_m1) {
          const TRef q = a.m1;
          const bool qring = q.mode == 0;
          const int qmask = qring ? q.lmask : -1;
  #pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int sidx = qring ? rslot[gq] : ri[gq];
            mkv[gq] = q.base[(long long)sidx * q.slot_stride + (((qring ? rpos[gq] * q.rate : 0) + q.off + rt[gq]) & qmask)];
          }
        }
        if (a.has_m2) {
          const TRef q = a.m2;
          const bool qring = q.mode == 0;
          const int qmask = qring ? q.lmask : -1;
  #pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int sidx = qring ? rslot[gq] : ri[gq];
            mkv[gq] *= q.base[(long long)sidx * q.slot_stride + (((qring ? rpos[gq] * q.rate : 0) + q.off + rt[gq]) & qmask)];
          }
        }
      }
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        const int co4 = EPF ? pf_co4 : n0 + (wn * RN + rn) * 32 + ec4 * 4;      // first of the lane's 4 channels
        const bool cok = co4 < Cout;
        const int cc = cok ? co4 : 0;
        float4 rv[4], bq;
        if constexpr (EPF) {
          bq = pf_bq;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) rv[gq] = pf_rv[gq];
        } else {
          // (2) residual + style vector (16-byte loads where the layout allows)
  #pragma unroll
          for (int gq = 0; gq < 4; ++gq) rv[gq] = f4zero();
          if (has_res) {
            const TRef rr = a.res;
            const bool rring = rr.mode == 0;
            const int rmask = rring ? rr.lmask : -1;
  #pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const int sidx = rring ? rslot[gq] : ri[gq];
              const int row = ((rring ? rpos[gq] * rr.rate : 0) + rr.off + rt[gq]) & rmask;
              const float* p = rr.base + (long long)sidx * rr.slot_stride + row * rr.C + cc;
              if (vec_ok && (rr.C & 3) == 0) rv[gq] = *reinterpret_cast<const float4*>(p);
              else { rv[gq].x = p[0]; rv[gq].y = cc + 1 < Cout ? p[1] : 0.f; rv[gq].z = cc + 2 < Cout ? p[2] : 0.f; rv[gq].w = cc + 3 < Cout ? p[3] : 0.f; }
            }
          }
          if (has_bvec) {
  #pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const float* bv = a.bvec + (long long)rslot[gq] * a.bvec_stride + cc;
              rv[gq].x += bv[0]; rv[gq].y += cc + 1 < Cout ? bv[1] : 0.f; rv[gq].z += cc + 2 < Cout ? bv[2] : 0.f; rv[gq].w += cc + 3 < Cout ? bv[3] : 0.f;
            }
          }
          bq = f4zero();
          if (a.bias) { bq.x = a.bias[cc]; bq.y = a.bias[cc + 1]; bq.z = a.bias[cc + 2]; bq.w = a.bias[cc + 3]; }   // bias is padded to Cout_pad
        }
        // (3) transpose the accumulator tile through the wave's LDS patch
#pragma unroll
        for (int e = 0; e < 16; ++e) patch[((e & 3) + 8 * (e >> 2) + 4 * lh) * EPI_LD + l31] = acc[rm][rn][e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float4 v[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) v[gq] = *reinterpret_cast<const float4*>(patch + (er + 8 * gq) * EPI_LD + ec4 * 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // (4) bias -> scale -> activation -> (+style vector, +residual) -> mask -> store
        int jj = 0, ocol = cc;
        if (shuf > 1) { jj = cc / Cq; ocol = cc - jj * Cq; }
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          float o[4] = {v[gq].x + bq.x, v[gq].y + bq.y, v[gq].z + bq.z, v[gq].w + bq.w};
          const float r4[4] = {rv[gq].x, rv[gq].y, rv[gq].z, rv[gq].w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float x = o[k] * oscale;
            if constexpr (ACT == ACT_LRELU) x = x > 0.f ? x : x * oslope;
            if constexpr (ACT == ACT_RELU) x = x > 0.f ? x : 0.f;
            if constexpr (ACT == ACT_GELU) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
            if constexpr (ACT == ACT_TANH) x = tanhf(x);
            o[k] = (x + r4[k]) * mkv[gq];
          }
          if (((okbits >> gq) & 1u) && cok) {
            float* ybase = ybase0 + (long long)(yring ? rslot[gq] : ri[gq]) * yss;
            const int yrow = ((yring ? rpos[gq] * yrate : 0) + yoff + rt[gq] * shuf + jj) & ymask;
            float* dst = ybase + yrow * yC + ocol;
            if (vec_ok) {
              *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
              if (y2base0) {
                float* dst2 = y2base0 + (dst - ybase0);
                *reinterpret_cast<float4*>(dst2) = make_float4(o[0] > 0.f ? o[0] : o[0] * y2slope, o[1] > 0.f ? o[1] : o[1] * y2slope,
                                                               o[2] > 0.f ? o[2] : o[2] * y2slope, o[3] > 0.f ? o[3] : o[3] * y2slope);
              }
            }
            else {
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                if (cc + k < Cout) {
                  int j2 = 0, oc2 = cc + k;
                  if (shuf > 1) { j2 = oc2 / Cq; oc2 -= j2 * Cq; }
                  ybase[(((yring ? rpos[gq] * yrate : 0) + yoff + rt[gq] * shuf + j2) & ymask) * yC + oc2] = o[k];
                }
              }
            }
          }
        }
      }
    }
  };
  switch (oact) {
    case ACT_LRELU: epilogue(std::integral_constant<int, ACT_LRELU>{}); break;
    case ACT_RELU: epilogue(std::integral_constant<int, ACT_RELU>{}); break;
    case ACT_GELU: epilogue(std::integral_constant<int, ACT_GELU>{}); break;
    case ACT_TANH: epilogue(std::integral_constant<int, ACT_TANH>{}); break;
    default: epilogue(std::integral_constant<int, ACT_NONE>{}); break;
  }
#ifdef CK_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CK_RSTAMP(4);
  if (a.dbg && tile == 1 && tid == 0) {
    a.dbg[0] = c_wait; a.dbg[1] = c_comp; a.dbg[2] = (unsigned long long)nks; a.dbg[3] = c_kend - c_begin; a.dbg[4] = __builtin_amdgcn_s_memtime() - c_kend; a.dbg[5] = __builtin_amdgcn_s_memtime() - c_begin; a.dbg[6] = __builtin_amdgcn_s_memrealtime() - c_rt0;
  }
#endif
#undef CK_CSTAMP
}

// Persistent launch: the grid is at most the number of co-resident blocks; every block walks the tile list
// tile = blockIdx.x, blockIdx.x + gridDim.x, ...  Tiles are ordered problem by problem, longest K first (the three
// resblock branches of a grouped launch have k = 11 / 7 / 3), n-tile fastest (neighbours share activation rows).
template <int TM, int TN, int WM, int WN, int WK, int KS>
__global__ __launch_bounds__(512, (ConvLds<TM, TN, WK, KS>::TOTAL * 4 > 80 * 1024) ? 2 : 4) void conv_mfma_kernel(const ConvGroup g) {
  __shared__ __attribute__((aligned(16))) float lds[ConvLds<TM, TN, WK, KS>::TOTAL];
  int gbuf = 0;
  const int S = g.ksplit;
  const int sfrom = g.split_from;                  // items [0, sfrom): whole tiles; then S slices per tile
  const int total = sfrom + (g.tile_start[3] - sfrom) * S;
#ifdef CK_STAMPS
  if (g.p[0].dbg && threadIdx.x == 0 && blockIdx.x < 512) g.p[0].dbg[32 + blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime();
#endif
  // Work items: round-robin over the grid, or - for persistent launches - the list launch_conv balanced on the host
  // (the three branches of a grouped resblock launch have k = 11 / 7 / 3, so equal tile counts are unequal work).
  const int* const assign = g.assign ? g.assign + (long long)blockIdx.x * g.assign_per : nullptr;
  int i = 0;
  int cur = assign ? assign[0] : (int)blockIdx.x;
  while (cur >= 0 && cur < total) {
    const int rs = cur - sfrom;
    const int tile = rs < 0 ? cur : sfrom + rs / S, kslice = rs < 0 ? 0 : rs % S;        // slices of a tile are adjacent: they finish together
    const int ns = rs < 0 ? 1 : S;
    const int p = (tile >= g.tile_start[1] ? 1 : 0) + (tile >= g.tile_start[2] ? 1 : 0);
    const int local = tile - g.tile_start[p];
    const int tn = g.tiles_n[p];
    const int mt = local / tn, nt = local - mt * tn;
    conv_tile<TM, TN, WM, WN, WK, KS>(g.p[g.order[p]], mt * TM, nt * TN, lds, gbuf, tile, kslice, ns,
                                            g.slab + (long long)(tile - sfrom) * S * (TM * TN), g.counters + tile, g.fenced);
    ++i;
    cur = assign ? (i < g.assign_per ? assign[i] : -1) : cur + (int)gridDim.x;
  }
#ifdef CK_STAMPS
  if (g.p[0].dbg && threadIdx.x == 0 && blockIdx.x < 512) g.p[0].dbg[32 + blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

static const int kTM[NUM_CFG] = {128, 64, 128, 32, 32, 64, 64, 128};
static const int kTN[NUM_CFG] = {64, 64, 32, 64, 32, 32, 64, 32};
int conv_cfg_tm(int cfg) { return kTM[cfg]; }
int conv_cfg_tn(int cfg) { return kTN[cfg]; }
static const int kKS[NUM_CFG] = {32, 32, 32, 64, 128, 32, 64, 64};
int conv_cfg_ks(int cfg) { return kKS[cfg]; }
bool conv_cfg_splitk(int cfg) { return kTM[cfg] == 32 || cfg == CFG_64x64 || cfg == CFG_128x64; }
const char* conv_cfg_name(int cfg) {
  static const char* names[NUM_CFG] = {
      "cnk::conv_mfma_kernel<128, 64, 2, 2, 1, 32>", "cnk::conv_mfma_kernel<64, 64, 2, 2, 1, 32>", "cnk::conv_mfma_kernel<128, 32, 4, 1, 1, 32>",
      "cnk::conv_mfma_kernel<32, 64, 1, 2, 2, 64>",  "cnk::conv_mfma_kernel<32, 32, 1, 1, 4, 128>", "cnk::conv_mfma_kernel<64, 32, 2, 1, 2, 32>",
      "cnk::conv_mfma_kernel<64, 64, 2, 2, 1, 64>",  "cnk::conv_mfma_kernel<128, 32, 4, 1, 1, 64>"};
  return cfg >= 0 && cfg < NUM_CFG ? names[cfg] : "?";
}

// Schedule of a persistent launch, computed on the host and cached per launch shape in device memory (a handful of
// shapes per model and device; never freed; guarded by a mutex):
//  * XCD-aware: workgroup b runs on XCD b % 8 and every XCD has its own 4 MB L2, so each XCD gets one contiguous
//    eighth of every problem's m-tiles (or, where the weights are the larger operand, of its n-tiles) - its slice of
//    the big operand then stays in that L2 instead of being fetched over the fabric by all eight (measured: L2 hit rate 78-85 % either way at
//    these sizes, so the step time does not move; it matters once a stage's activations outgrow one L2);
//  * balanced: within an XCD, longest-processing-time-first over its blocks (the three branches of a grouped resblock
//    launch have k = 11 / 7 / 3); cost of an item = its K-steps + a constant for prologue/epilogue.
static const int* balanced_assignment(const ConvGroup& g, int grid, int KS, int* per_out) {
  struct Key { int v[14]; bool operator<(const Key& o) const { return memcmp(v, o.v, sizeof(v)) < 0; } };
  static std::map<Key, std::pair<const int*, int>> cache;
  static std::mutex cache_mu;                       // host threads driving different devices / stream-sets share it
  std::lock_guard<std::mutex> lock(cache_mu);
  constexpr int NX = 8;
  int nks[3];
  for (int q = 0; q < 3; ++q) { const ConvArgs& a = g.p[g.order[q]]; nks[q] = a.ktaps * ((a.Cin_pad + KS - 1) / KS); }
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);                     // the cached lists live in that device's memory
  Key k = {{grid, g.tile_start[1], g.tile_start[2], g.tile_start[3], nks[0], nks[1], nks[2], KS, g.tiles_n[0], g.tiles_n[1], g.tiles_n[2], dev_id, g.ksplit, g.split_from}};
  auto it = cache.find(k);
  if (it != cache.end()) { *per_out = it->second.second; return it->second.first; }
  const int ntiles = g.tile_start[3], S = std::max(1, g.ksplit), sfrom = g.ksplit > 1 ? g.split_from : ntiles;
  const int total = sfrom + (ntiles - sfrom) * S;      // work items: whole tiles, then the K slices of the tail tiles
  std::vector<std::vector<int>> lists(grid);
  typedef std::pair<long long, int> Bin;
  std::priority_queue<Bin, std::vector<Bin>, std::greater<Bin>> heap[NX];
  const bool by_xcd = grid >= NX * 2 && grid % NX == 0;
  for (int b = 0; b < grid; ++b) heap[by_xcd ? b % NX : 0].push({0, b});
  for (int it = 0; it < total; ++it) {
    const int item = it < sfrom ? it : sfrom + (it - sfrom) / S;      // the tile of this work item
    const int q = (item >= g.tile_start[1] ? 1 : 0) + (item >= g.tile_start[2] ? 1 : 0);
    const int local = item - g.tile_start[q], tn = g.tiles_n[q];
    const int n_mt = (g.tile_start[q + 1] - g.tile_start[q]) / tn, mt = local / tn, nt = local - mt * tn;
    // split along the axis whose operand is the larger one: activations (rows) for the streaming layers, weights
    // (columns) for small-M x huge-K*N layers such as ups[0] (0.5 MB of rows against 67 MB of weights)
    const ConvArgs& a = g.p[g.order[q]];
    const bool by_cols = (long long)a.ktaps * a.Cin * a.Cout > (long long)a.n * a.T * a.Cin && tn >= NX;
    const int x = !by_xcd ? 0 : by_cols ? std::min(NX - 1, (int)((long long)nt * NX / tn))
                                        : std::min(NX - 1, (int)((long long)mt * NX / std::max(n_mt, 1)));
    Bin top = heap[x].top(); heap[x].pop();
    lists[top.second].push_back(it);
    heap[x].push({top.first + (it < sfrom ? nks[q] : (nks[q] + S - 1) / S) + 3, top.second});
  }
  // The two blocks that share a CU would otherwise walk equal-length tiles in lockstep and reach their epilogues -
  // where the matrix pipe idles - together: every other block of an XCD runs its list shortest-first.
  for (int b = 0; b < grid; ++b) if ((b / NX) & 1) std::reverse(lists[b].begin(), lists[b].end());
  size_t per = 1;
  for (auto& l : lists) per = std::max(per, l.size() + 1);
  std::vector<int> flat((size_t)grid * per, -1);
  for (int b = 0; b < grid; ++b) std::copy(lists[b].begin(), lists[b].end(), flat.begin() + (size_t)b * per);
  int* dev = nullptr;
  if (hipMalloc(&dev, flat.size() * sizeof(int)) != hipSuccess) { *per_out = 0; return nullptr; }
  (void)hipMemcpy(dev, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice);
  cache[k] = {dev, (int)per};
  *per_out = (int)per;
  return dev;
}

template <int TM, int TN, int WM, int WN, int WK, int KS>
static void launch_one(const ConvGroup& gin, int num_cu, hipStream_t st) {
  constexpr int lds_bytes = ConvLds<TM, TN, WK, KS>::TOTAL * 4;
  const int per_cu = lds_bytes > 80 * 1024 ? 1 : 2;
  ConvGroup g = gin;
  g.assign = nullptr; g.assign_per = 0;
  if (g.ksplit <= 1) { g.ksplit = 1; g.split_from = g.tile_start[3]; }
  if (g.split_from > g.tile_start[3]) g.split_from = g.tile_start[3];
  int grid = g.split_from + (g.tile_start[3] - g.split_from) * g.ksplit;
  if (grid > num_cu * per_cu) {     // persistent blocks
    grid = num_cu * per_cu;
    g.assign = balanced_assignment(g, grid, KS, &g.assign_per);
  }
  hipLaunchKernelGGL((conv_mfma_kernel<TM, TN, WM, WN, WK, KS>), dim3(grid), dim3(512), 0, st, g);
}

static void launch_conv_cfg(const ConvGroup& g, int cfg, int num_cu, hipStream_t st) {
  switch (cfg) {
    case CFG_128x64: launch_one<128, 64, 2, 2, 1, 32>(g, num_cu, st); break;
    case CFG_64x64: launch_one<64, 64, 2, 2, 1, 32>(g, num_cu, st); break;
    case CFG_128x32: launch_one<128, 32, 4, 1, 1, 32>(g, num_cu, st); break;
    case CFG_32x64_K2: launch_one<32, 64, 1, 2, 2, 64>(g, num_cu, st); break;
    case CFG_32x32_K4: launch_one<32, 32, 1, 1, 4, 128>(g, num_cu, st); break;
    case CFG_64x32_K2: launch_one<64, 32, 2, 1, 2, 32>(g, num_cu, st); break;
    case CFG_64x64_KS64: launch_one<64, 64, 2, 2, 1, 64>(g, num_cu, st); break;
    case CFG_128x32_KS64: launch_one<128, 32, 4, 1, 1, 64>(g, num_cu, st); break;
    default: break;
  }
}

void launch_conv(const ConvGroup& gin, int nprob, int cfg, hipStream_t st, int num_cu) {
  ConvGroup g = gin;
  if (g.ksplit < 1 || !g.slab || !g.counters || !conv_cfg_splitk(cfg)) g.ksplit = 1;
  const int TM = kTM[cfg], TN = kTN[cfg];
  // longest-K problem first
  int idx[3] = {0, 1, 2};
  for (int i = 0; i < nprob; ++i)
    for (int j = i + 1; j < nprob; ++j) {
      const long long ki = (long long)g.p[idx[i]].ktaps * g.p[idx[i]].Cin_pad, kj = (long long)g.p[idx[j]].ktaps * g.p[idx[j]].Cin_pad;
      if (kj > ki) { int t = idx[i]; idx[i] = idx[j]; idx[j] = t; }
    }
  int acc = 0;
  for (int q = 0; q < 3; ++q) {
    g.tile_start[q] = acc;
    if (q < nprob) {
      const ConvArgs& a = g.p[idx[q]];
      const int tm = (a.n * a.T + TM - 1) / TM, tn = (a.Cout + TN - 1) / TN;
      g.order[q] = idx[q]; g.tiles_n[q] = tn;
      acc += tm * tn;
    } else { g.order[q] = 0; g.tiles_n[q] = 1; }
  }
  g.tile_start[3] = acc;
  if (acc == 0) return;
  launch_conv_cfg(g, cfg, num_cu, st);
}

}  // namespace cnk
