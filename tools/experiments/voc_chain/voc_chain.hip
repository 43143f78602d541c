// voc_chain: the HiFi-GAN vocoder step of a SMALL stream-set (slots x frames <= 16 mel rows: one to four streams of 80 ms chunks)
// as ONE persistent launch - HifiGanGenerator.forward, modules/vocoder/hifigan/hifigan_causal.py:314-333, with
// CausalUpsampleBlock3 + CausalPixelShuffle1d (:171-212) and ResBlock1 (:217-244).
//
// Why: at one stream the step was 25 split-K conv launches + the fused C = 32 stage, each a chain of dependent memory round trips
// (kernel arguments -> slot table -> frame counter -> window -> cold weights -> split-K hand-off -> residual -> store) of 13-15 us
// for a microsecond of arithmetic: 0.40 ms per chunk against a weight-stream floor of 120 MB / 6 TB/s = 20 us.  Here the step is a
// list of PHASES (one conv layer of all three branches each) walked by a resident grid:
//   * a phase is cut into jobs = (branch, row tile(s), column group) that the workgroups deal round-robin; a job is worked on by
//     the 8 waves of a workgroup as NCT column tiles x KS K-slices (NCT * KS = 8), partial tiles meet in LDS in slice order;
//   * the slot table and the frame counters are read ONCE per launch; every wave requests the first VC_RING weight fragments of its
//     next job BEFORE it waits for the previous phase (weights do not depend on it), so what a phase waits for is one poll, one
//     window gather, its MFMAs and the acknowledgement of its stores;
//   * activations cross workgroups (on different XCDs, whose L2s are not coherent) as agent-scope write-through stores and sc1
//     loads, ordered by every storing wave's s_waitcnt vmcnt(0), the workgroup's barrier and one arrival on a counter that counts
//     for ever (MI355X_MICROARCH.md, "Valid forms"; the same hand-off as decoder_mega.hip and conv_mfma's split-K) - no fence;
//   * the MFMAs are issued transposed (weights as the first operand): a lane's accumulator holds 4 consecutive output channels of
//     ONE row, so the epilogue's loads and stores are 16 bytes wide; pixel shuffle is the store's address;
//   * LeakyReLU(mean of the branches) (hifigan_causal.py:324-331) is formed by the consumer's gather ((v0 + v1) + v2, / 3,
//     LeakyReLU: mean_act_kernel's operations) - no phase of its own - and appended to the stage's activated-mean ring by the
//     job of column group 0; the mel chunk reaches its ring the same way.
// Rings and their contents are those of the two-launch plan without the activated twins (LeakyReLU is applied in the gather).
// Waits are bounded (kernels.h, SpinGuard).  Forward progress needs the whole grid resident: the host launches at most
// voc_chain_max_grid() workgroups and serialises the chain launches of a context with an event.
#include <algorithm>

#include "kernels.h"

namespace cnk {
namespace {

typedef float vf4 __attribute__((ext_vector_type(4)));
typedef unsigned vu4 __attribute__((ext_vector_type(4)));
typedef const vf4 __attribute__((address_space(1)))* vc_gcf4;
typedef vf4 __attribute__((address_space(1)))* vc_gf4;
typedef const float __attribute__((address_space(1)))* vc_gcf1;
typedef const int __attribute__((address_space(4)))* vc_cci;
#define VC_AS4(T, p) (*(const T __attribute__((address_space(4)))*)(p))

constexpr int VC_THREADS = 512, VC_WAVES = 8;
constexpr int VC_RING = 20;           // weight fragments (16 bytes per lane each) a wave keeps in flight
constexpr int VC_HDR = 192;           // LDS header: slot[16] | pos[16] | misc[32] | tab[64] | spare
constexpr int VC_SC1 = 16;
constexpr int VC_POST_ROWS = 256;     // conv_post rows per job

__device__ __forceinline__ __amdgpu_buffer_rsrc_t vc_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ vf4 vc_xload4(__amdgpu_buffer_rsrc_t r, int float_off) {
  const vu4 u = __builtin_amdgcn_raw_buffer_load_b128(r, float_off * 4, 0, VC_SC1);
  return (vf4){__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3])};
}
__device__ __forceinline__ void vc_xstore4(__amdgpu_buffer_rsrc_t r, int float_off, const vf4 v) {
  __builtin_amdgcn_raw_buffer_store_b128((vu4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, r, float_off * 4, 0, VC_SC1);
}
__device__ __forceinline__ vf4 vc_wload4(const float* p) { return *(vc_gcf4)(p); }
// 16 bytes through a per-lane address, bypassing the reading CU's L1 (two 8-byte agent-scope loads: global_load_dwordx2 ... sc1)
typedef const unsigned long long __attribute__((address_space(1)))* vc_gcu64;
__device__ __forceinline__ vf4 vc_gload4(const float* p) {
  const unsigned long long a = __hip_atomic_load((vc_gcu64)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load((vc_gcu64)(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (vf4){__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32))};
}

// float offset of row t of (batch index i / slot) from the tensor's base
template <class TR>
__device__ __forceinline__ int vc_off(const TR& r, int i, int slot, int pos, int t) {
  if (r.mode == 0) return (int)((long long)slot * r.slot_stride) + (int)(((unsigned)pos * (unsigned)r.rate + (unsigned)(r.off + t)) & (unsigned)r.lmask) * r.C;
  return (int)((long long)i * r.slot_stride) + (r.off + t) * r.C;
}
__device__ __forceinline__ vf4 vc_lrelu(vf4 v, float s) {
  v[0] = v[0] > 0.f ? v[0] : v[0] * s; v[1] = v[1] > 0.f ? v[1] : v[1] * s; v[2] = v[2] > 0.f ? v[2] : v[2] * s; v[3] = v[3] > 0.f ? v[3] : v[3] * s;
  return v;
}

// ---- phase hand-off: one arrival counter per phase (64 bytes apart, zero at launch, zeroed again by the launch's last workgroup);
// a workgroup arrives when ITS stores of the phase are out (at once if it had no job), and waits - only in front of a job - until all
// G workgroups have arrived for the phase before
__device__ __forceinline__ void vc_arrive(unsigned* bar) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every wave's write-through stores have been acknowledged ...
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ... before the workgroup's one arrival
}
__device__ __forceinline__ void vc_wait(unsigned* bar, const unsigned target, unsigned* guard) {
  if (threadIdx.x == 0) {
    SpinGuard sg;
    while ((int)(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (spin_expired(sg, guard, WAIT_VOC_PHASE)) break;
    }
  }
  __syncthreads();
}

// geometry of a job's rows
struct VCTile { int i0, nseg, t0, Tr, nrows, halo, seglen, WR; };

// developer stamps (CONAN_VC_STAMPS=1): per phase the latest "wait passed", "window gathered", "K loops done" and "stores issued" over
// all workgroups, on the 100 MHz clock
__device__ __forceinline__ void vc_stamp(unsigned long long* dbg, const int phase, const int what) {
  if (dbg && threadIdx.x == 0) dbg[((long long)blockIdx.x * VC_MAX_PHASES + phase) * 4 + what] = __builtin_amdgcn_s_memrealtime();      // (the workgroup's own words: its LAST job of the phase)
}

// K loop of one wave: column tile ct, K groups [g_lo, g_hi) of k * KQ, NRT row tiles.  bw holds fragments g_lo .. g_lo + VC_RING - 1.
template <int NRT>
__device__ __forceinline__ void vc_kloop(vf4 (&acc)[NRT], vf4 (&bw)[VC_RING], const float* __restrict__ wl, const int g_lo, const int g_hi, const int KQ,
                                         const int tstep, const float* (&abase)[NRT]) {
  int q = g_lo % KQ, j = g_lo / KQ;
  vf4 acc2 = {0.f, 0.f, 0.f, 0.f};
  vf4 af[NRT];
#pragma unroll
  for (int r = 0; r < NRT; ++r) af[r] = *reinterpret_cast<const vf4*>(abase[r] + j * tstep + q * 16);
  for (int G0 = g_lo; G0 < g_hi; G0 += VC_RING) {
#pragma unroll
    for (int u = 0; u < VC_RING; ++u) {
      const bool live = G0 + u < g_hi;
      if (++q == KQ) { q = 0; ++j; }
      const bool nlive = G0 + u + 1 < g_hi;
      vf4 afn[NRT];
#pragma unroll
      for (int r = 0; r < NRT; ++r) afn[r] = *reinterpret_cast<const vf4*>(abase[r] + (nlive ? j * tstep + q * 16 : 0));
      if (live) {
        if constexpr (NRT == 1) {
          // one row tile: its MFMAs would form ONE dependent chain (40-cycle latency against a 32-cycle issue interval) - two
          // interleaved chains, summed behind the loop (the same K sum in a different association)
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[u][0], af[0][0], acc[0], 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[u][1], af[0][1], acc2, 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[u][2], af[0][2], acc[0], 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[u][3], af[0][3], acc2, 0, 0, 0);
        } else {
#pragma unroll
          for (int r = 0; r < NRT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[u][0], af[r][0], acc[r], 0, 0, 0);
#pragma unroll
          for (int r = 0; r < NRT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[u][1], af[r][1], acc[r], 0, 0, 0);
#pragma unroll
          for (int r = 0; r < NRT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[u][2], af[r][2], acc[r], 0, 0, 0);
#pragma unroll
          for (int r = 0; r < NRT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[u][3], af[r][3], acc[r], 0, 0, 0);
        }
      }
      // refill this ring slot - UNCONDITIONALLY (past the wave's range: its last group once more, an L1 hit): behind a branch hipcc
      // cannot count the load and opens every round of the ring with s_waitcnt vmcnt(0), i.e. with the full latency of the refills
      { const int gn = G0 + u + VC_RING < g_hi ? G0 + u + VC_RING : g_hi - 1; bw[u] = vc_wload4(wl + (long long)(gn - g_lo) * 256); }
#pragma unroll
      for (int r = 0; r < NRT; ++r) af[r] = afn[r];
    }
  }
  if constexpr (NRT == 1) acc[0] += acc2;
}

// One conv job: gather the window, K loops, slice sums, epilogue.  `first` = the workgroup's first job of the phase: the wait for the
// previous phase sits behind everything that does not depend on it - the wave's first VC_RING weight fragments and its bias are
// requested, the addresses of the window's first round of loads and of the epilogue's residual / output rows are worked out.
// NSRC = 3: the new rows of the input are leaky_relu(mean of three raw branch outputs) (rings of one geometry).
template <int NRT, int NSRC>
__device__ __forceinline__ void vc_conv_job(const VCPhase* __restrict__ ph, const int job, float* __restrict__ lds, const VCIO& io, const bool first,
                                            unsigned* bar, const unsigned wait_target, unsigned* guard, unsigned long long* dbg, const int phase) {
  int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  int* const s_slot = reinterpret_cast<int*>(lds);
  int* const s_pos = s_slot + 16;
  int* const tab = s_slot + 64;
  float* const win = lds + VC_HDR;
  vc_cci hd = (vc_cci)(ph);
  const int n = hd[2], T = hd[3], NCT = hd[5], KS = hd[6], tps = hd[7], spt = hd[8], tiles = hd[9], ncg = hd[10];
  const unsigned magic_c4 = (unsigned)hd[12];
  // ---- decode the job: problem (branch), row tile, column group
  const int per_prob = tiles * ncg;
  const int pi = job / per_prob, rem = job - pi * per_prob;
  const int tile = rem / ncg, cg = rem - tile * ncg;
  const VCProb* pp = &ph->p[pi];
  const auto& P = VC_AS4(VCProb, pp);
  const int Cin = P.Cin, k = P.k, dil = P.dil, KQ = P.KQ;
  VCTile tl;
  if (spt == 0) { tl.i0 = tile / tps; const int rt = tile - tl.i0 * tps; tl.t0 = rt * 16 * NRT; tl.Tr = min(16 * NRT, T - tl.t0); tl.nseg = 1; }
  else { tl.i0 = tile * spt; tl.nseg = min(spt, n - tl.i0); tl.t0 = 0; tl.Tr = T; }
  tl.nrows = tl.nseg * tl.Tr; tl.halo = (k - 1) * dil; tl.seglen = tl.halo + tl.Tr; tl.WR = tl.nseg * tl.seglen;
  const int LDX = Cin + 8, C4 = Cin >> 2;
  float* const red = win + tl.WR * LDX;        // partial tiles of the K slices, behind the window
  // ---- this wave's column tile and K slice; its first weight fragments are requested now
  const int cti = wv / KS, ks = wv - cti * KS;
  const int ct = cg * NCT + cti;
  const int GT = k * KQ;
  const int g_lo = (GT * ks) / KS, g_hi = (GT * (ks + 1)) / KS;
  const float* wl = P.w + ((long long)ct * (k + 1) * KQ + g_lo) * 256 + lane * 4;
  vf4 bw[VC_RING];
#pragma unroll
  for (int u = 0; u < VC_RING; ++u) bw[u] = vc_wload4(wl + (long long)u * 256);
  // ---- the output tiles this wave finishes: with one K slice its own column tile's NRT row tiles, else tile f = wv (+ 8, ..)
  const int sr = P.shuffle_r, Cq = P.Cq;
  const int ffc = KS == 1 ? cti : wv / NRT, fr0 = KS == 1 ? 0 : wv - (wv / NRT) * NRT;
  constexpr int NFIN = NRT;                    // (KS > 1: only [0] is used)
  int yoff[NFIN], roff[NFIN];
  bool fon[NFIN];
  const int fct = cg * NCT + ffc, c0 = fct * 16 + 4 * lg;
#pragma unroll
  for (int q = 0; q < NFIN; ++q) {
    const int r = KS == 1 ? q : fr0;
    const int row = r * 16 + lr;
    fon[q] = (KS == 1 || (q == 0 && wv < NCT * NRT)) && fct < P.ncts && row < tl.nrows && c0 < P.Cout;
    const int rowc = fon[q] ? row : 0;
    const int seg = rowc / tl.Tr, tr = rowc - seg * tl.Tr;
    const int i = tl.i0 + seg, slot = s_slot[i], pos = s_pos[i], tau = tl.t0 + tr;
    int ot = tau, oc = c0;
    if (sr > 1) { const int jj = c0 / Cq; ot = tau * sr + jj; oc = c0 - jj * Cq; }
    yoff[q] = vc_off(P.y, i, slot, pos, ot) + oc;
    roff[q] = P.has_res ? vc_off(P.res, i, slot, pos, tau) + c0 : 0;
  }
  const vf4 pbias = (fct < P.ncts && c0 < P.Cout) ? vc_wload4(P.bias + c0) : (vf4){0.f, 0.f, 0.f, 0.f};
  // ---- the window's first round of loads: (source, offset) per element
  const int total = tl.WR * C4;
  constexpr int U = NSRC == 1 ? 9 : 3;
  // element e -> window row w, 16-byte column c4; time tau (negative: earlier steps); source offset in floats
  auto plan = [&](const int e, int& meta, int& off) __attribute__((always_inline)) {
    const int ec = e < total ? e : 0;
    const int w = (int)__umulhi((unsigned)ec, magic_c4), c4 = ec - w * C4;
    int seg = 0;
    for (int s2 = 1; s2 < tl.nseg; ++s2) seg += (w >= s2 * tl.seglen) ? 1 : 0;
    const int tau = tl.t0 - tl.halo + (w - seg * tl.seglen);
    const int i = tl.i0 + seg, slot = s_slot[i], pos = s_pos[i];
    off = (tau < 0 ? vc_off(P.xhist, i, slot, pos, tau) : vc_off(P.xnew[0], i, slot, pos, tau)) + c4 * 4;
    meta = w | (c4 << 10) | (seg << 20) | (tau < 0 ? (1 << 28) : 0) | (e < total ? (1 << 29) : 0);
  };
  int gmeta[U], goff[U];
#pragma unroll
  for (int u = 0; u < U; ++u) plan(tid + VC_THREADS * u, gmeta[u], goff[u]);
  const float* const xh_base = P.xhist.base;
  const float* const x0_base = P.io_in ? io.mel : P.xnew[0].base;
  const float* const x1_base = NSRC > 1 ? P.xnew[1].base : xh_base;
  const float* const x2_base = NSRC > 2 ? P.xnew[2].base : xh_base;
  const float* const res_base = P.has_res ? P.res.base : xh_base;
  const __amdgpu_buffer_rsrc_t ry = vc_rsrc(P.y.base);
  const __amdgpu_buffer_rsrc_t rr = vc_rsrc(P.has_res ? P.res.base : P.y.base);
  const bool keep = P.store_new && cg == 0;
  float* const tapn = (keep && P.tap_new >= 0) ? io.tap[P.tap_new] : nullptr;
  const bool form = P.store_new && (NSRC > 1 || P.mean_slope != 1.0f);      // the new rows are leaky_relu(mean of the sources)
  // =================================================================== everything below depends on the phase before
  if (first) vc_wait(bar, wait_target, guard);
  else __syncthreads();                      // every wave is done with the previous job's LDS
  vc_stamp(dbg, phase, 0);
  if (tid < 16 * NRT) {
    const int seg = tid / tl.Tr, tr = tid - seg * tl.Tr;
    tab[tid] = tid < tl.nrows ? seg * tl.seglen + tr : 0;
  }
  // ---- gather the window [WR][Cin] (LeakyReLU / branch mean applied on the way)
  vf4 rres[NFIN];
  for (int e0 = 0; e0 < total; e0 += VC_THREADS * U) {
    vf4 v0[U], v1[NSRC > 1 ? U : 1], v2[NSRC > 2 ? U : 1];
    if (e0 > 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) plan(e0 + tid + VC_THREADS * u, gmeta[u], goff[u]);
    }
    // (unconditional, straight-line loads: a row of earlier steps reads its ring through the same instruction as a new row its
    // source - per-lane base - and, with three sources, reads it three times; behind branches every load waits for the one before)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool hist = (gmeta[u] >> 28) & 1;
      const float* const bh = xh_base + goff[u];
      v0[u] = vc_gload4(hist ? bh : x0_base + goff[u]);
      if constexpr (NSRC > 1) v1[u] = vc_gload4(hist ? bh : x1_base + goff[u]);
      if constexpr (NSRC > 2) v2[u] = vc_gload4(hist ? bh : x2_base + goff[u]);
    }
    if (e0 == 0) {      // the residual rows of this wave's output tiles: behind the window's loads, in front of the K loop
#pragma unroll
      for (int q = 0; q < NFIN; ++q) rres[q] = vc_gload4((P.has_res && fon[q]) ? res_base + roff[q] : xh_base + goff[0]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!((gmeta[u] >> 29) & 1)) continue;
      const int w = gmeta[u] & 1023, c4 = (gmeta[u] >> 10) & 1023;
      const bool fresh = !((gmeta[u] >> 28) & 1);
      vf4 v = v0[u];
      if (fresh && form) {
        // leaky_relu(mean of the branch outputs): (v0 + v1) + v2, true division, LeakyReLU - mean_act_kernel's operations
        if constexpr (NSRC > 1) v += v1[u];
        if constexpr (NSRC > 2) v += v2[u];
        if constexpr (NSRC > 1) { const float dn = (float)NSRC; v[0] /= dn; v[1] /= dn; v[2] /= dn; v[3] /= dn; }
        v = vc_lrelu(v, P.mean_slope);
      }
      if (fresh && keep) {      // the rows formed here go to the ring they are history of in later steps (plain: nobody reads them in this launch)
        const int seg = (gmeta[u] >> 20) & 255;
        const int i = tl.i0 + seg, slot = s_slot[i], pos = s_pos[i];
        const int tau = tl.t0 - tl.halo + (w - seg * tl.seglen);
        *(vc_gf4)(P.xhist.base + vc_off(P.xhist, i, slot, pos, tau) + c4 * 4) = v;
        if (tapn) *(vc_gf4)(tapn + ((long long)i * T + tau) * Cin + c4 * 4) = v;
      }
      if (P.in_lrelu) v = vc_lrelu(v, P.in_slope);
      *reinterpret_cast<vf4*>(win + w * LDX + c4 * 4) = v;
    }
  }
  __syncthreads();
  vc_stamp(dbg, phase, 1);
  // ---- K loop
  vf4 acc[NRT];
#pragma unroll
  for (int r = 0; r < NRT; ++r) acc[r] = (vf4){0.f, 0.f, 0.f, 0.f};
  {
    const float* abase[NRT];
#pragma unroll
    for (int r = 0; r < NRT; ++r) abase[r] = win + tab[r * 16 + lr] * LDX + 4 * lg;
    if (ct < P.ncts) vc_kloop<NRT>(acc, bw, wl, g_lo, g_hi, KQ, dil * LDX, abase);
  }
  vc_stamp(dbg, phase, 2);
  // ---- slice sums through LDS, in slice order
  if (KS > 1) {
#pragma unroll
    for (int r = 0; r < NRT; ++r) *reinterpret_cast<vf4*>(red + ((wv * NRT + r) * 64 + lane) * 4) = acc[r];
    __syncthreads();
  }
  // ---- epilogue: lane (lg, lr) holds channels 4 lg .. 4 lg + 3 of row lr of its output tile
  float* const tapy = P.tap >= 0 ? io.tap[P.tap] : nullptr;
  const int Cy = P.y.C;
  auto tap_store = [&](const int r, const vf4 v) __attribute__((always_inline)) {      // (taps: a developer / test path)
    const int row = r * 16 + lr;
    const int seg = row / tl.Tr, tr = row - seg * tl.Tr;
    const int i = tl.i0 + seg, tau = tl.t0 + tr;
    int ot = tau, oc = c0;
    if (sr > 1) { const int jj = c0 / Cq; ot = tau * sr + jj; oc = c0 - jj * Cq; }
    *(vc_gf4)(tapy + ((long long)i * (T * sr) + ot) * Cy + oc) = v;
  };
  if (KS == 1) {        // a wave owns its column tile's whole K range: it finishes its own row tiles
#pragma unroll
    for (int q = 0; q < NFIN; ++q) {
      if (!fon[q]) continue;
      vf4 v = acc[q] + pbias;
      if (P.out_act == ACT_LRELU) v = vc_lrelu(v, P.out_slope);
      if (P.has_res) v += rres[q];
      vc_xstore4(ry, yoff[q], v);
      if (tapy) tap_store(q, v);
    }
  } else {
    if (fon[0]) {
      vf4 v = *reinterpret_cast<const vf4*>(red + (((ffc * KS) * NRT + fr0) * 64 + lane) * 4);
      for (int s2 = 1; s2 < KS; ++s2) v += *reinterpret_cast<const vf4*>(red + (((ffc * KS + s2) * NRT + fr0) * 64 + lane) * 4);
      v += pbias;
      if (P.out_act == ACT_LRELU) v = vc_lrelu(v, P.out_slope);
      if (P.has_res) v += rres[0];
      vc_xstore4(ry, yoff[0], v);
      if (tapy) tap_store(fr0, v);
    }
    for (int f = wv + VC_WAVES; f < NCT * NRT; f += VC_WAVES) {      // (more output tiles than waves: the rest, operands fetched here)
      const int fc = f / NRT, r = f - fc * NRT;
      const int fct2 = cg * NCT + fc, row = r * 16 + lr, c2 = fct2 * 16 + 4 * lg;
      if (fct2 >= P.ncts || row >= tl.nrows || c2 >= P.Cout) continue;
      vf4 v = *reinterpret_cast<const vf4*>(red + (((fc * KS) * NRT + r) * 64 + lane) * 4);
      for (int s2 = 1; s2 < KS; ++s2) v += *reinterpret_cast<const vf4*>(red + (((fc * KS + s2) * NRT + r) * 64 + lane) * 4);
      const int seg = row / tl.Tr, tr = row - seg * tl.Tr;
      const int i = tl.i0 + seg, slot = s_slot[i], pos = s_pos[i], tau = tl.t0 + tr;
      v += vc_wload4(P.bias + c2);
      if (P.out_act == ACT_LRELU) v = vc_lrelu(v, P.out_slope);
      if (P.has_res) v += vc_xload4(rr, vc_off(P.res, i, slot, pos, tau) + c2);
      int ot = tau, oc = c2;
      if (sr > 1) { const int jj = c2 / Cq; ot = tau * sr + jj; oc = c2 - jj * Cq; }
      vc_xstore4(ry, vc_off(P.y, i, slot, pos, ot) + oc, v);
      if (tapy) *(vc_gf4)(tapy + ((long long)i * (T * sr) + ot) * Cy + oc) = v;
    }
  }
  vc_stamp(dbg, phase, 3);
}

// conv_post (CausalConv1d(C -> 1, k) + tanh, hifigan_causal.py:331-333) on leaky_relu(mean of the last stage's branches): a job is
// VC_POST_ROWS output samples of one slot, one thread per sample (conv_post_kernel's arithmetic)
__device__ __forceinline__ void vc_post_job(const VCPhase* __restrict__ ph, const int job, float* __restrict__ lds, const VCIO& io, const bool first,
                                            unsigned* bar, const unsigned wait_target, unsigned* guard) {
  const int tid = threadIdx.x;
  int* const s_slot = reinterpret_cast<int*>(lds);
  int* const s_pos = s_slot + 16;
  float* const win = lds + VC_HDR;
  vc_cci hd = (vc_cci)(ph);
  const int T = hd[3], tps = hd[7], k = hd[13];
  const unsigned magic_c4 = (unsigned)hd[12];
  const float bpost = __int_as_float(hd[14]);
  const auto& P = VC_AS4(VCProb, &ph->p[0]);
  const float* wpost = VC_AS4(float*, &ph->wpost);
  const int C = P.Cin, C4 = C >> 2, ld = C + 4;
  const int i = job / tps, t0 = (job - i * tps) * VC_POST_ROWS;
  const int rows = min(VC_POST_ROWS, T - t0) + k - 1;
  float* const sw = win + (VC_POST_ROWS + k - 1) * ld;
  if (first) vc_wait(bar, wait_target, guard);
  else __syncthreads();
  for (int e = tid; e < k * C; e += VC_THREADS) sw[e] = *(vc_gcf1)(wpost + e);
  const int slot = s_slot[i], pos = s_pos[i];
  const __amdgpu_buffer_rsrc_t rh = vc_rsrc(P.xhist.base);
  const int nsrc = P.nsrc;
  const __amdgpu_buffer_rsrc_t r0 = vc_rsrc(P.xnew[0].base);
  const __amdgpu_buffer_rsrc_t r1 = vc_rsrc(nsrc > 1 ? P.xnew[1].base : P.xhist.base);
  const __amdgpu_buffer_rsrc_t r2 = vc_rsrc(nsrc > 2 ? P.xnew[2].base : P.xhist.base);
  float* const tapn = P.tap_new >= 0 ? io.tap[P.tap_new] : nullptr;
  const int total = rows * C4;
  constexpr int U = 4;
  for (int e0 = 0; e0 < total; e0 += VC_THREADS * U) {
    vf4 v0[U], v1[U], v2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + tid + VC_THREADS * u, ec = e < total ? e : 0;
      const int w = (int)__umulhi((unsigned)ec, magic_c4), c4 = ec - w * C4;
      const int tau = t0 + w - (k - 1);
      v1[u] = v2[u] = (vf4){0.f, 0.f, 0.f, 0.f};
      if (tau < 0) v0[u] = vc_xload4(rh, vc_off(P.xhist, i, slot, pos, tau) + c4 * 4);
      else {
        v0[u] = vc_xload4(r0, vc_off(P.xnew[0], i, slot, pos, tau) + c4 * 4);
        if (nsrc > 1) v1[u] = vc_xload4(r1, vc_off(P.xnew[1], i, slot, pos, tau) + c4 * 4);
        if (nsrc > 2) v2[u] = vc_xload4(r2, vc_off(P.xnew[2], i, slot, pos, tau) + c4 * 4);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + tid + VC_THREADS * u;
      if (e >= total) continue;
      const int w = (int)__umulhi((unsigned)e, magic_c4), c4 = e - w * C4;
      const int tau = t0 + w - (k - 1);
      vf4 v = v0[u];
      if (tau >= 0) {
        if (nsrc > 1) v += v1[u];
        if (nsrc > 2) v += v2[u];
        if (nsrc > 1) { const float dn = (float)nsrc; v[0] /= dn; v[1] /= dn; v[2] /= dn; v[3] /= dn; }
        v = vc_lrelu(v, P.mean_slope);
        if (w >= k - 1) {      // this job's own rows -> the activated-mean ring (history of later steps)
          *(vc_gf4)(P.xhist.base + vc_off(P.xhist, i, slot, pos, tau) + c4 * 4) = v;
          if (tapn) *(vc_gf4)(tapn + ((long long)i * T + tau) * C + c4 * 4) = v;
        }
      }
      *reinterpret_cast<vf4*>(win + w * ld + c4 * 4) = v;
    }
  }
  __syncthreads();
  const int t = t0 + tid;
  if (tid < VC_POST_ROWS && t < T) {
    float acc = 0.f;
    for (int j = 0; j < k; ++j) {
      const vf4* xr = reinterpret_cast<const vf4*>(win + (tid + j) * ld);
      const vf4* wr = reinterpret_cast<const vf4*>(sw + j * C);
      for (int c0 = 0; c0 < C4; c0 += 8) {
        vf4 x4[8], w4[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int c = c0 + u < C4 ? c0 + u : c0; x4[u] = xr[c]; w4[u] = wr[c]; }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (c0 + u < C4) { acc += x4[u][0] * w4[u][0]; acc += x4[u][1] * w4[u][1]; acc += x4[u][2] * w4[u][2]; acc += x4[u][3] * w4[u][3]; }
      }
    }
    acc += bpost;
    const long long o = (long long)i * T + t;
    if (io.pre) io.pre[o] = acc;
    io.wav[o] = tanhf(acc);
  }
}

__global__ __launch_bounds__(VC_THREADS, 2) void voc_chain_kernel(const VCPhase* __restrict__ prog, const int nphases, const int* __restrict__ slots,
                                                                  int* __restrict__ pos, const int n, const int adv, unsigned* __restrict__ bar,
                                                                  unsigned* __restrict__ guard, const VCIO io, unsigned long long* __restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = (int)blockIdx.x, G = (int)gridDim.x;
  {   // slot table and frame counters of the launch's streams: read once
    int* const s_slot = reinterpret_cast<int*>(lds);
    if (threadIdx.x < 16) {
      const int q = threadIdx.x < n ? threadIdx.x : 0;
      const int sl = slots[q];
      s_slot[threadIdx.x] = sl; s_slot[16 + threadIdx.x] = pos[sl];
    }
  }
  __syncthreads();
  if (dbg && threadIdx.x == 0) dbg[((long long)G * VC_MAX_PHASES) * 4 + b] = __builtin_amdgcn_s_memrealtime();
  for (int p = 0; p < nphases; ++p) {
    const VCPhase* ph = prog + p;
    vc_cci hd = (vc_cci)(ph);
    const int type = hd[0], NRT = hd[4], njobs = hd[11];
    const bool nsrc3 = VC_AS4(VCProb, &ph->p[0]).nsrc == 3;
    unsigned* const dep = bar + (p > 0 ? (p - 1) * 16 : 0);      // arrivals of the phase before: all G of them
    bool first = p > 0;
    // (jobs are dealt from a workgroup index that rotates with the phase: the workgroups that were busy last are not the first again)
    int j0 = b - (int)((unsigned)(p * 37) % (unsigned)G);
    if (j0 < 0) j0 += G;
    for (int job = j0; job < njobs; job += G) {
      if (type == 1) vc_post_job(ph, job, lds, io, first, dep, (unsigned)G, guard);
      else if (nsrc3) {
        if (NRT == 1) vc_conv_job<1, 3>(ph, job, lds, io, first, dep, (unsigned)G, guard, dbg, p);
        else if (NRT == 2) vc_conv_job<2, 3>(ph, job, lds, io, first, dep, (unsigned)G, guard, dbg, p);
        else vc_conv_job<4, 3>(ph, job, lds, io, first, dep, (unsigned)G, guard, dbg, p);
      } else {
        if (NRT == 1) vc_conv_job<1, 1>(ph, job, lds, io, first, dep, (unsigned)G, guard, dbg, p);
        else if (NRT == 2) vc_conv_job<2, 1>(ph, job, lds, io, first, dep, (unsigned)G, guard, dbg, p);
        else vc_conv_job<4, 1>(ph, job, lds, io, first, dep, (unsigned)G, guard, dbg, p);
      }
      first = false;
    }
    if (p + 1 < nphases) vc_arrive(bar + p * 16);
  }
  // the last workgroup to finish advances the frame counters (every workgroup read them at the start of the launch) and re-arms
  // the phase counters: by then every workgroup is past its last wait
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* const s_last = reinterpret_cast<int*>(lds) + 32;      // (header word: no static LDS in front of the dynamic region - its base stays 16-byte aligned)
  unsigned* const fin = bar + VC_MAX_PHASES * 16;
  if (threadIdx.x == 0) *s_last = __hip_atomic_fetch_add(fin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)G - 1u;
  __syncthreads();
  if (*s_last) {
    if (dbg && threadIdx.x == 0) dbg[((long long)G * VC_MAX_PHASES) * 4 + G] = __builtin_amdgcn_s_memrealtime();
    for (int q = threadIdx.x; q < n; q += VC_THREADS) pos[slots[q]] += adv;
    for (int q = threadIdx.x; q <= VC_MAX_PHASES; q += VC_THREADS) __hip_atomic_store(bar + q * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// conv_mfma's packed weight -> fragment-major [ncts][k + 1][KQ][64][4] (the zero tap and the slack stay zero)
__global__ void voc_chain_repack_kernel(float* __restrict__ dst, const float* __restrict__ w, const int Cout, const int Cin, const int Cin_alloc, const int k) {
  const int KQ = Cin / 16, ncts = (Cout + 15) / 16;
  const long long total = (long long)ncts * k * KQ * 256;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int s = (int)(e & 3), lane = (int)((e >> 2) & 63);
    long long g = e >> 8;
    const int q = (int)(g % KQ); g /= KQ;
    const int j = (int)(g % k); const int ct = (int)(g / k);
    const int ci = q * 16 + 4 * (lane >> 4) + s, col = ct * 16 + (lane & 15);
    float v = 0.f;
    if (col < Cout) v = w[((((long long)(col / 64) * k + j) * (Cin_alloc / 4) + ci / 4) * 64 + (col % 64)) * 4 + (ci & 3)];
    dst[(((long long)ct * (k + 1) + j) * KQ + q) * 256 + lane * 4 + s] = v;
  }
}

}  // namespace

size_t voc_chain_weight_floats(int Cout, int Cin, int k) {
  const int KQ = Cin / 16, ncts = (Cout + 15) / 16;
  return ((size_t)ncts * (k + 1) * KQ + 2 * VC_RING) * 256;
}

void launch_voc_chain_repack(float* dst, const float* w, int Cout, int Cout_pad, int Cin, int Cin_alloc, int k, hipStream_t st) {
  (void)Cout_pad;
  hipLaunchKernelGGL(voc_chain_repack_kernel, dim3(512), dim3(256), 0, st, dst, w, Cout, Cin, Cin_alloc, k);
}

int voc_chain_max_grid(int lds_bytes, int num_cu) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, voc_chain_kernel, VC_THREADS, (size_t)lds_bytes) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return nb * num_cu;
}

void launch_voc_chain(const VCLaunch& l, hipStream_t st) {
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(voc_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024); attr = true; }
  hipLaunchKernelGGL(voc_chain_kernel, dim3(l.grid), dim3(VC_THREADS), l.lds_bytes, st, l.prog, l.nphases, l.slots, l.pos, l.n, l.adv, l.bar, l.guard, l.io, l.dbg);
}

}  // namespace cnk
