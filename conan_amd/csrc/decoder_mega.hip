// decoder_mega: the Conan decoder step (content embedding, content_proj, the two-layer prosody aligner, the uv / f0
// predictor and pitch embedding, the four causal conv blocks, post conv and mel_out - modules/Conan/Conan.py:140-198,
// :324-351, :584-589) as ONE persistent launch.
//
// Why: as ~38 dependent launches the step is latency-bound alone (0.55 ms at 64 streams) and, in the pipelined chunk step,
// every one of those launches has to find room for its workgroups on CUs that the vocoder's persistent kernels hold - the
// decoder stage then takes 1.8 ms per chunk and IS the pipeline's bottleneck (conan_step_timeline).  Here the workgroups
// are placed once (256 threads, <= 80 VGPRs, ~34 KB of LDS: they fit on a CU beside a vocoder workgroup).
//
// Structure: every operator of the step works on rows (stream, frame) and needs, of the operators before it, only rows of
// the SAME streams.  A job is therefore one 16-row tile (4 streams x 4 frames) taken through the whole operator list by a
// GROUP of workgroups: for a conv / linear operator every member gathers the tile's input window into its LDS once
// (LayerNorm applied there), then computes its share of the 64-column output strips (member s takes strips s, s + GS, ...:
// with groups laid out member-fastest, strip s always runs on XCD s % 8 and its weights stay in that XCD's L2); row-wise
// operators (LayerNorm, cross attention, pitch head, embedding) split the tile's 16 rows over the members.  Dependent
// operators are separated by a barrier of the GROUP only (arrival counter + sc1 polling: 1.6 us for 8 members against
// 2.3 us for a 128-workgroup grid barrier) - groups never wait for each other until the step's last operator, the
// frame-counter advance, which sits behind one grid barrier.  When a stream's frames of the step do not all fall into one
// tile (16 % frames != 0: the ragged last chunk) the program runs with ONE group and tiles as jobs in sequence - no,
// simpler and rarer: such steps keep the separate launches (run_mega refuses them).
// With a single row tile in the step (<= 4 streams) the one group is the whole grid and a strip is 16 columns wide, its K
// loop split over the 4 waves of a workgroup (rowconv's <1,1,4> build).
//
// Coherence inside the launch: the per-XCD L2s are not coherent and group members sit on different XCDs, so activations
// go out as agent-scope write-through stores and come in through sc1 loads (rowops.h, COH = true: 0 stale reads and no
// fence - tools/experiments/uncached_barrier.hip); weights, the per-utterance caches and the frame counters are read-only
// until the advance and stay plain.
// Forward progress: the barriers need every workgroup resident.  The grid's workgroups can share CUs with each other (4 per
// CU by LDS) and with the vocoder's, and nothing resident ever waits for this kernel, so it is always eventually placed
// in full ("decoder megakernel").
#include <algorithm>
#include <atomic>

#include "kernels.h"
#include "rowops.h"

namespace cnk {

#ifndef MG_POLL_SLEEP
#define MG_POLL_SLEEP 1
#endif
typedef const int __attribute__((address_space(4)))* mg_cci;
#define MG_AS4(T, p) (*(const T __attribute__((address_space(4)))*)(p))

// The operators read their arguments where they use them, straight out of the program (device memory, constant for the
// launch) through the scalar cache: references into the constant address space.  Small argument blocks are copied:
template <typename T>
__device__ __forceinline__ T mg_args(const void* p) {
  static_assert(sizeof(T) % 4 == 0, "argument structs are made of 32-bit words");
  union { T v; int w[sizeof(T) / 4]; } u;
  mg_cci s = (mg_cci)(p);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 4); ++i) u.w[i] = s[i];
  return u.v;
}

// The operators read their arguments through the scalar cache, and every operator's block is cold the first time it is touched: a
// chain of 4-5 dependent misses (~0.4 us each) at the head of every operator.  One scalar load per 64-byte line of the NEXT operator's
// block, issued in front of the barrier that ends this one, takes them off the critical path.
__device__ __forceinline__ void mg_prefetch_args(const MegaOp* op) {
  mg_cci p = (mg_cci)(op);
  int x = 0;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(MegaOp) / 4); i += 16) x ^= p[i];
  asm volatile("" ::"s"(x));
}

// arrival counter + sc1 polling; `target` is the count at which every participant has arrived
__device__ __forceinline__ void mg_barrier(unsigned* ctr, const unsigned target, unsigned* guard) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every wave's (write-through) stores have left ...
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ... before the workgroup's one arrival
    SpinGuard sg;
    while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(MG_POLL_SLEEP);
      if (spin_expired(sg, guard, WAIT_MEGA_BARRIER)) break;        // (bounded: kernels.h, SpinGuard)
    }
  }
  __syncthreads();
}

// xcd mode (CM = 2): every participant sits on ONE XCD, whose L2 is the point of coherence.  Atomics would still execute at the memory
// side (a fabric round trip each); flags do not: member `rank` stores the barrier's epoch into its own word of one 128-byte line
// (plain store -> L2), wave 0 of every member polls all P words with L1-bypassing loads (L2 hits) until none is older.  Epochs grow
// from launch to launch (the host passes the launch's base), nothing is ever cleared.
__device__ __forceinline__ void mg_barrier_xcd(unsigned* flags, const unsigned epoch, const int rank, const int P, unsigned* guard) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every wave's stores have reached the L2 ...
  __syncthreads();
  if (threadIdx.x < 64) {
    typedef unsigned __attribute__((address_space(1)))* gu32;
    if (threadIdx.x == 0) *(volatile gu32)(flags + rank) = epoch;      // ... before the member's flag
    const int lane = threadIdx.x;
    SpinGuard sg;
    for (;;) {
      const unsigned v = lane < P ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
      if (__all((int)(v - epoch) >= 0)) break;
      if (spin_expired(sg, guard, WAIT_MEGA_FLAGS)) break;
    }
  }
  __syncthreads();
}

// gbar: one arrival counter per group, 64 bytes apart (zero at launch, zeroed again by the kernel); bar: the grid's (counts for ever)
// OCC = waves per SIMD the register allocation is bounded for: 6 (80 registers: a wave fits beside two waves of every vocoder
// kernel, the f32 ResBlock passes' 216 included; the operators' bodies then spill ~35 registers to scratch - reloaded once per
// operator, outside the K loops) for exact-f32 stream-sets; 4 (128 registers, no spills) for bf16-limb stream-sets, whose
// resident kernels (resblock_limb <= 192, conv_limb's streaming shapes <= 192) leave 2 x 192 + 128 = 512 - tests/test_kernel_resources.py.
// CM = 2 (xcd mode, a single row tile in the step): the launch is one workgroup per CU; the workgroups that find themselves on the
// XCD of workgroup 0 (HW_REG_XCC_ID - read, not assumed) form the ONE group that walks the program, the others leave at once.
// xs: [0] election word, [16 .. 48) two sets of per-XCD arrival counters, [64 .. 128) the xcd group's barrier flags; CM = 3:
// [128 .. 192) the groups' XCC masks, [256 ..) 32 flag words per group.
template <int OCC, int CM>
__global__ __launch_bounds__(256, OCC) void decoder_mega_kernel(const MegaOp* __restrict__ prog, const int nops, const int njobs, const int GS_,
                                                              const int* __restrict__ slots, const int* __restrict__ pos, const int n, const int T,
                                                              unsigned* __restrict__ gbar, unsigned* __restrict__ bar, const unsigned bar_base,
                                                              unsigned long long* __restrict__ dbg, unsigned* __restrict__ guard,
                                                              unsigned* __restrict__ xs, const unsigned xseq, const unsigned xdec_base) {
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  ro::RowTab& tab = *reinterpret_cast<ro::RowTab*>(lds_all);
  float* const lds = lds_all + ro::ROWTAB_FLOATS;
  int b = (int)blockIdx.x;
#ifdef MG_PRIO
  __builtin_amdgcn_s_setprio(MG_PRIO);      // developer build
#endif
  int GS = GS_;
  unsigned xepoch = xseq << 12;      // (barrier epochs: up to 4096 per launch, growing from launch to launch)
  if constexpr (CM == 2) {
    // Quorum election.  Every workgroup registers with its XCD (arrival rank r on that XCD's counter).  The XCD whose count first
    // reaches 32 - or, after ~4 us, the first XCD with at least 8 - is claimed (one compare-and-swap on the election word, which
    // carries the launch's sequence number, the XCD and the member count P); its arrivals 0 .. P - 1 are the group, everybody else
    // leaves.  Nothing waits for workgroups that have not started: a CU held by another launch's spinning workgroups (another
    // stream-set's xcd group occupies a whole XCD; workgroups are bound to XCDs round-robin) only makes THAT XCD lose the election.
    // Counters: two sets of 16, used alternately; the group's rank 0 zeroes the other set at the end (its last users - the launch
    // before - are all gone).  The election word goes from (xseq - 1) << 12 ("done") to xseq << 12 | P << 4 | xcc and back to "done".
    int* const sh = reinterpret_cast<int*>(lds_all);
    if (threadIdx.x == 0) {
      const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;      // HW_REG_XCC_ID[3:0]
      unsigned* const cnt = xs + 16 + (xseq & 1u) * 16;
      const unsigned r = __hip_atomic_fetch_add(cnt + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned idle = ((xseq - 1u) & 0xfffffu) << 12, tag = (xseq & 0xfffffu) << 12;
      if (r + 1u == 32u) {       // this arrival completes the XCD: claim it
        unsigned exp = idle;
        __hip_atomic_compare_exchange_strong(xs, &exp, tag | (32u << 4) | xcc, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      unsigned v;
      SpinGuard sg;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (((v = __hip_atomic_load(xs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & ~0xfffu) != tag) {
        __builtin_amdgcn_s_sleep(MG_POLL_SLEEP);
        if (r == 0u && __builtin_amdgcn_s_memrealtime() - t0 > 400ull) {      // the XCD's first arrival: after 4 us a smaller group will do
          const unsigned c = __hip_atomic_load(cnt + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (c >= 8u) {
            unsigned exp = idle;
            __hip_atomic_compare_exchange_strong(xs, &exp, tag | ((c < 32u ? c : 32u) << 4) | xcc, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        if (spin_expired(sg, guard, WAIT_MEGA_ELECT)) break;
      }
      const unsigned P = (v >> 4) & 255u;
      const bool mine = (v & ~0xfffu) == tag && (v & 15u) == xcc && r < P;
      sh[0] = mine ? (int)r : -1; sh[1] = (int)P;
    }
    __syncthreads();
    const int rank = sh[0], P = sh[1];
    __syncthreads();
    if (rank < 0 || P <= 0) return;
    b = rank; GS = P;
    {   // pull the program into this XCD's L2 (the operators read their arguments through the scalar cache where they use them: the
        // first touch of every operator's block would otherwise be a chain of misses to memory)
      const int words = nops * (int)(sizeof(MegaOp) / 4);
      float keep = 0.f;
      for (int e = (b * 256 + (int)threadIdx.x) * 32; e < words; e += GS * 256 * 32) keep += ro::ldw1(reinterpret_cast<const float*>(prog) + e);
      ro::mg_keep(keep);
    }
  }
  // groups, this workgroup's group and member index.  CM = 3 (several row tiles): groups are laid out GROUP-fastest - workgroup b is
  // member b / NG of group b % NG - so that with the dispatcher's round-robin over the 8 XCDs and a group count that is a multiple of
  // 8 (the host pads it) all members of a group land on ONE XCD.  That is checked, not assumed: every member ORs its XCC id into the
  // group's mask word; one bit set -> the group's activations stay in that XCD's L2 (plain stores, flag barriers: tab.l2 = 1),
  // otherwise the group keeps the agent-scope protocol (write-through stores, counter barriers).
  const int NG = CM == 2 ? 1 : (int)gridDim.x / GS;
  const bool gfast = CM == 3 && !(xdec_base & 2u);      // (xdec_base bit 1: member-fastest - member s of every group on XCD s)
  const int g = gfast ? b % NG : b / GS, sb = gfast ? b / NG : b - g * GS;
  if (threadIdx.x == 0) tab.l2 = CM == 2 ? 1 : 0;
  // developer stamps (CONAN_MEGA_STAMPS=1): workgroup 0 notes the 100 MHz clock at the start and behind every operator (+ barrier)
  if (dbg && b == 0 && threadIdx.x == 0) dbg[0] = __builtin_amdgcn_s_memrealtime();
  unsigned gtarget = 0;
  int l2 = CM == 2 ? 1 : 0;
  if constexpr (CM == 3) {
    if (g < njobs) {
      if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;      // HW_REG_XCC_ID[3:0]
        __hip_atomic_fetch_or(xs + 128 + g, 1u << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      gtarget += (unsigned)GS; mg_barrier(gbar + g * 16, gtarget, guard);
      if (threadIdx.x == 0) {
        const unsigned m = __hip_atomic_load(xs + 128 + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tab.l2 = ((m & (m - 1u)) == 0u && xdec_base == 0u) ? 1 : 0;      // (xdec_base != 0: developer switch CONAN_MEGA_NOL2 - the agent-scope protocol everywhere)
      }
      __syncthreads();
      l2 = tab.l2;
      if (dbg && sb == 0 && threadIdx.x == 0) dbg[kMegaDbgGroupWords + g] = 0x100u | (unsigned)l2 | (__hip_atomic_load(xs + 128 + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 16);
    }
  }
  for (int job = g; job < njobs; job += NG) {
    // row table of the job's tile: stream / frame / slot / frame counter of its rows (the same for every operator)
    __syncthreads();
    ro::rowtab_setup(tab, slots, pos, n, T, job * ro::RC_TM);
    __syncthreads();
    for (int o = 0; o < nops; ++o) {
      const MegaOp* op = prog + o;
      mg_cci hdr = (mg_cci)(op);
      const int type = hdr[0], nbx = hdr[1], barrier = hdr[3];
      switch (type) {
        case MOP_RC111: {      // 64-column strips
          const auto& a = MG_AS4(RowConvArgs, &op->u);
          if (sb < nbx) {      // (members without a strip skip the gather as well)
            float4 bw[8];
            // (the 8-deep ring of this form held across the gather costs the 128-register build 20 spilled registers: L2 warm-up only)
            if (dbg && b == 0 && threadIdx.x == 0) dbg[128 + o * 4 + 0] = __builtin_amdgcn_s_memrealtime();
            if constexpr (OCC < 6) {
              // (warm-up loads IN FRONT of the gather: vector-memory loads return in order, so the gather's L2 hits then wait for these
              // misses - ~2 us in the stage - but the K loop runs out of a warm L2: 0.434 ms per 64-stream step against 0.451 with the
              // warm-up behind the gather's loads and 0.459 with a single warm-up line)
              const ro::Warm10 warm = ro::mg_wwarm_all<1>(a, sb);
              ro::mg_stage<true, CM>(a, tab, lds, sb, nbx < GS ? nbx : GS);
              ro::mg_keep(warm);
            } else {
              const float warm = ro::mg_wwarm<1>(a, sb);
              ro::mg_stage<false, CM>(a, tab, lds, sb, nbx < GS ? nbx : GS);
              ro::mg_keep(warm);
            }
            if (dbg && b == 0 && threadIdx.x == 0) dbg[128 + o * 4 + 1] = __builtin_amdgcn_s_memrealtime();
            for (int bx = sb; bx < nbx; bx += GS) ro::mg_strip<1, false, CM, (OCC < 6)>(a, tab, bx, lds, bw);
            if (dbg && b == 0 && threadIdx.x == 0) dbg[128 + o * 4 + 2] = __builtin_amdgcn_s_memrealtime();
          }
        } break;
        case MOP_RC114: {      // 16-column strips, K split over the waves (a single tile in the step)
          const auto& a = MG_AS4(RowConvArgs, &op->u);
          if (sb < nbx) {
            float4 bw[4];
            if (dbg && b == 0 && threadIdx.x == 0) dbg[128 + o * 4 + 0] = __builtin_amdgcn_s_memrealtime();
            if constexpr (OCC < 6) {
              ro::mg_wpre<4>(a, sb, bw);
              ro::mg_stage<true, CM>(a, tab, lds, sb, nbx < GS ? nbx : GS);
              if (dbg && b == 0 && threadIdx.x == 0) dbg[128 + o * 4 + 1] = __builtin_amdgcn_s_memrealtime();
              ro::mg_strip<4, true, CM, true>(a, tab, sb, lds, bw);
              for (int bx = sb + GS; bx < nbx; bx += GS) ro::mg_strip<4, false, CM, true>(a, tab, bx, lds, bw);
            } else {
              const float warm = ro::mg_wwarm<4>(a, sb);
              ro::mg_stage<false, CM>(a, tab, lds, sb, nbx < GS ? nbx : GS);
              ro::mg_keep(warm);
              if (dbg && b == 0 && threadIdx.x == 0) dbg[128 + o * 4 + 1] = __builtin_amdgcn_s_memrealtime();
              for (int bx = sb; bx < nbx; bx += GS) ro::mg_strip<4, false, CM, false>(a, tab, bx, lds, bw);
            }
            if (dbg && b == 0 && threadIdx.x == 0) dbg[128 + o * 4 + 2] = __builtin_amdgcn_s_memrealtime();
          }
        } break;
        case MOP_FFN: {        // LayerNorm -> 1x1 -> activation -> 1x1 partial sums, the hidden columns split over the members
          const auto& a = MG_AS4(RowConvArgs, &op->u);
          if constexpr (CM == 2) {
            // xcd mode: the group's size is only known at run time - the hidden columns are split over Cout / 64 VIRTUAL members
            // (one 64-column strip each: 32 for the aligner's 2048-wide feed-forward), dealt to the workgroups that are there
            // (the LayerNorm prologue's row stores are owned by the members that stage: only those that run the loop - a feed-forward
            // narrower than 64 x the group size leaves the others without a virtual member)
            const int VG = a.Cout >> 6;
            for (int v = sb; v < VG; v += GS) {
              ro::mg_stage<(OCC < 6), CM>(a, tab, lds, sb, GS < VG ? GS : VG);
              ro::mg_ffn<(OCC < 6), CM>(a, tab, v, VG, lds);
            }
          } else {
            ro::mg_stage<(OCC < 6), CM>(a, tab, lds, sb, GS);
            ro::mg_ffn<(OCC < 6), CM>(a, tab, sb, GS, lds);
          }
        } break;
        case MOP_ROWLIN: {
          const auto& a = MG_AS4(RowConvArgs, &op->u);
          for (int bx = sb; bx < nbx; bx += GS) ro::mg_rowlin_strip<CM>(a, tab, bx, lds);
        } break;
        case MOP_LN: {
          const auto& a = MG_AS4(LNArgs, &op->u);
          for (int r = sb * 4 + (int)(threadIdx.x >> 6); r < ro::RC_TM; r += GS * 4) ro::mg_layernorm_row<CM, (OCC < 6)>(a, tab, r);      // 16 rows over GS members x 4 waves
        } break;
        case MOP_XATTN: {
          const auto& a = MG_AS4(XAttnArgs, &op->u);
          for (int r0 = sb * 2; r0 < ro::RC_TM; r0 += GS * 2) { __syncthreads(); ro::mg_xattn_rows<CM, (OCC < 6)>(a, tab, r0, lds); }
        } break;
        case MOP_PITCH: {
          const auto& a = MG_AS4(PitchHeadArgs, &op->u);
          const float f0 = __int_as_float(hdr[4]), f1 = __int_as_float(hdr[5]);
          for (int r = sb * 4 + (int)(threadIdx.x >> 6); r < ro::RC_TM; r += GS * 4) ro::mg_pitch_row<CM>(a, tab, f0, f1, r);
        } break;
        case MOP_EMBED: {
          const auto& a = MG_AS4(EmbedArgs, &op->u);
          for (int r = sb * 4 + (int)(threadIdx.x >> 6); r < ro::RC_TM; r += GS * 4) ro::mg_embed_row<CM>(a, tab, r);
        } break;
        case MOP_COPY32: {
          if (job == g) {         // once per launch: spread over the grid
            const MegaCopy a = mg_args<MegaCopy>(&op->u);
            for (long long e = (long long)b * 256 + threadIdx.x; e < a.n; e += (long long)(CM == 2 ? GS : (int)gridDim.x) * 256) a.dst[e] = a.src[e];
          }
        } break;
        default: break;           // (MOP_ADVANCE: behind the grid barrier below)
      }
      if (o + 1 < nops) mg_prefetch_args(prog + o + 1);
      if (barrier && type != MOP_ADVANCE) {
        if constexpr (CM == 2) mg_barrier_xcd(xs + 64, ++xepoch, sb, GS, guard);
        else if (CM == 3 && l2) mg_barrier_xcd(xs + 256 + g * 32, ++xepoch, sb, GS, guard);
        else { gtarget += (unsigned)GS; mg_barrier(gbar + g * 16, gtarget, guard); }
      }
      if (dbg && b == 0 && threadIdx.x == 0 && job == g) dbg[1 + o] = __builtin_amdgcn_s_memrealtime();
    }
  }
  // every job is done: advance the frame counters (the operators above read them), re-arm the group counters
  if constexpr (CM == 2) mg_barrier_xcd(xs + 64, ++xepoch, sb, GS, guard);
  else mg_barrier(bar, bar_base + (unsigned)gridDim.x, guard);
  if (b == 0) {
    for (int o = 0; o < nops; ++o) {
      mg_cci hdr = (mg_cci)(prog + o);
      if (hdr[0] == MOP_ADVANCE) {
        const MegaAdvance a = mg_args<MegaAdvance>(&(prog + o)->u);
        for (int q = threadIdx.x; q < a.n; q += 256) a.pos[a.slots ? a.slots[q] : q] += a.delta;
      }
    }
    if constexpr (CM == 2) {      // the other counter set (the launch before's: all its workgroups are gone) and the election word
      if (threadIdx.x < 16) __hip_atomic_store(xs + 16 + ((xseq + 1u) & 1u) * 16 + threadIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (threadIdx.x == 0) __hip_atomic_store(xs, (xseq & 0xfffffu) << 12, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    else for (int q = threadIdx.x; q < NG; q += 256) {
      __hip_atomic_store(gbar + q * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (CM == 3) __hip_atomic_store(xs + 128 + q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the groups' XCC masks
    }
  }
  if (dbg && b == 0 && threadIdx.x == 0) dbg[1 + nops] = __builtin_amdgcn_s_memrealtime();
}

int decoder_mega_lds_floats(const MegaOp& op, int rc_lds_floats) {
  switch (op.type) {
    case MOP_RC111: case MOP_RC114: case MOP_ROWLIN: case MOP_FFN: return rc_lds_floats + ro::ROWTAB_FLOATS;
    case MOP_XATTN: return 2 * ro::XA2_LDS_FLOATS + ro::ROWTAB_FLOATS;
    default: return ro::ROWTAB_FLOATS;
  }
}

// workgroups of the megakernel that one CU can hold at once with `lds_bytes` of dynamic LDS (its barriers need the whole grid resident)
int decoder_mega_blocks_per_cu(int lds_bytes, bool wide_regs) {
  int nb = 0;
  const hipError_t e = wide_regs ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, decoder_mega_kernel<4, 3>, 256, (size_t)lds_bytes)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, decoder_mega_kernel<6, 3>, 256, (size_t)lds_bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
  return nb;
}

void launch_decoder_mega(const MegaLaunch& m, hipStream_t st) {
  if (m.xcd) {
    // (the attribute is per device: one bit per device id, like conv_limb's and rowconv's)
    static std::atomic<unsigned long long> attr_devs{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_mega_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      attr_devs.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((decoder_mega_kernel<4, 2>), dim3(m.groups * m.group_size), dim3(256), m.lds_bytes, st, m.prog, m.nops, m.njobs, m.group_size, m.slots, m.pos, m.n, m.T,
                       m.gbar, m.bar, m.bar_base, m.dbg, m.guard, m.xs, m.xseq, m.xdec_base);
  } else if (m.wide_regs)
    hipLaunchKernelGGL((decoder_mega_kernel<4, 3>), dim3(m.groups * m.group_size), dim3(256), m.lds_bytes, st, m.prog, m.nops, m.njobs, m.group_size, m.slots, m.pos, m.n, m.T,
                       m.gbar, m.bar, m.bar_base, m.dbg, m.guard, m.xs, m.xseq, m.xdec_base);
  else
    hipLaunchKernelGGL((decoder_mega_kernel<6, 3>), dim3(m.groups * m.group_size), dim3(256), m.lds_bytes, st, m.prog, m.nops, m.njobs, m.group_size, m.slots, m.pos, m.n, m.T,
                       m.gbar, m.bar, m.bar_base, m.dbg, m.guard, m.xs, m.xseq, m.xdec_base);
}

}  // namespace cnk
