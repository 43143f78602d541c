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

// gbar: one arrival counter per group, 64 bytes apart (zero at launch, zeroed again by the kernel); bar: the grid's (counts for ever)
// OCC = waves per SIMD the register allocation is bounded for: 6 (80 registers: a wave fits beside two waves of every vocoder
// kernel, the f32 ResBlock passes' 216 included; the operators' bodies then spill ~35 registers to scratch - reloaded once per
// operator, outside the K loops) for exact-f32 stream-sets; 4 (128 registers, no spills) for bf16-limb stream-sets, whose
// resident kernels (resblock_limb <= 192, conv_limb's streaming shapes <= 192) leave 2 x 192 + 128 = 512 - tests/test_kernel_resources.py.
template <int OCC>
__global__ __launch_bounds__(256, OCC) void decoder_mega_kernel(const MegaOp* __restrict__ prog, const int nops, const int njobs, const int GS,
                                                              const int* __restrict__ slots, const int* __restrict__ pos, const int n, const int T,
                                                              unsigned* __restrict__ gbar, unsigned* __restrict__ bar, const unsigned bar_base,
                                                              unsigned long long* __restrict__ dbg, unsigned* __restrict__ guard) {
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  ro::RowTab& tab = *reinterpret_cast<ro::RowTab*>(lds_all);
  float* const lds = lds_all + ro::ROWTAB_FLOATS;
  const int b = (int)blockIdx.x;
#ifdef MG_PRIO
  __builtin_amdgcn_s_setprio(MG_PRIO);      // developer build
#endif
  const int NG = (int)gridDim.x / GS, g = b / GS, sb = b - g * GS;      // groups, this workgroup's group and member index
  // developer stamps (CONAN_MEGA_STAMPS=1): workgroup 0 notes the 100 MHz clock at the start and behind every operator (+ barrier)
  if (dbg && b == 0 && threadIdx.x == 0) dbg[0] = __builtin_amdgcn_s_memrealtime();
  unsigned gtarget = 0;
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
            const float warm = ro::mg_wwarm<1>(a, sb);
            ro::mg_stage<(OCC < 6)>(a, tab, lds, sb, nbx < GS ? nbx : GS);
            ro::mg_keep(warm);
            for (int bx = sb; bx < nbx; bx += GS) ro::mg_strip<1, false>(a, tab, bx, lds, bw);
          }
        } break;
        case MOP_RC114: {      // 16-column strips, K split over the waves (a single tile in the step)
          const auto& a = MG_AS4(RowConvArgs, &op->u);
          if (sb < nbx) {
            float4 bw[4];
            const float warm = ro::mg_wwarm<4>(a, sb);
            ro::mg_stage<(OCC < 6)>(a, tab, lds, sb, nbx < GS ? nbx : GS);
            ro::mg_keep(warm);
            for (int bx = sb; bx < nbx; bx += GS) ro::mg_strip<4, false>(a, tab, bx, lds, bw);
          }
        } break;
        case MOP_FFN: {        // LayerNorm -> 1x1 -> activation -> 1x1 partial sums, the hidden columns split over the members
          const auto& a = MG_AS4(RowConvArgs, &op->u);
          ro::mg_stage<(OCC < 6)>(a, tab, lds, sb, GS);
          ro::mg_ffn<(OCC < 6)>(a, tab, sb, GS, lds);
        } break;
        case MOP_ROWLIN: {
          const auto& a = MG_AS4(RowConvArgs, &op->u);
          for (int bx = sb; bx < nbx; bx += GS) ro::mg_rowlin_strip(a, tab, bx, lds);
        } break;
        case MOP_LN: {
          const auto& a = MG_AS4(LNArgs, &op->u);
          for (int r = sb * 4 + (int)(threadIdx.x >> 6); r < ro::RC_TM; r += GS * 4) ro::mg_layernorm_row(a, tab, r);      // 16 rows over GS members x 4 waves
        } break;
        case MOP_XATTN: {
          const auto& a = MG_AS4(XAttnArgs, &op->u);
          for (int r0 = sb * 2; r0 < ro::RC_TM; r0 += GS * 2) { __syncthreads(); ro::mg_xattn_rows(a, tab, r0, lds); }
        } break;
        case MOP_PITCH: {
          const auto& a = MG_AS4(PitchHeadArgs, &op->u);
          const float f0 = __int_as_float(hdr[4]), f1 = __int_as_float(hdr[5]);
          for (int r = sb * 4 + (int)(threadIdx.x >> 6); r < ro::RC_TM; r += GS * 4) ro::mg_pitch_row(a, tab, f0, f1, r);
        } break;
        case MOP_EMBED: {
          const auto& a = MG_AS4(EmbedArgs, &op->u);
          for (int r = sb * 4 + (int)(threadIdx.x >> 6); r < ro::RC_TM; r += GS * 4) ro::mg_embed_row(a, tab, r);
        } break;
        case MOP_COPY32: {
          if (job == g) {         // once per launch: spread over the grid
            const MegaCopy a = mg_args<MegaCopy>(&op->u);
            for (long long e = (long long)b * 256 + threadIdx.x; e < a.n; e += (long long)gridDim.x * 256) a.dst[e] = a.src[e];
          }
        } break;
        default: break;           // (MOP_ADVANCE: behind the grid barrier below)
      }
      if (barrier && type != MOP_ADVANCE) { gtarget += (unsigned)GS; mg_barrier(gbar + g * 16, gtarget, guard); }
      if (dbg && b == 0 && threadIdx.x == 0 && job == g) dbg[1 + o] = __builtin_amdgcn_s_memrealtime();
    }
  }
  // every job is done: advance the frame counters (the operators above read them), re-arm the group counters
  mg_barrier(bar, bar_base + (unsigned)gridDim.x, guard);
  if (b == 0) {
    for (int o = 0; o < nops; ++o) {
      mg_cci hdr = (mg_cci)(prog + o);
      if (hdr[0] == MOP_ADVANCE) {
        const MegaAdvance a = mg_args<MegaAdvance>(&(prog + o)->u);
        for (int q = threadIdx.x; q < a.n; q += 256) a.pos[a.slots ? a.slots[q] : q] += a.delta;
      }
    }
    for (int q = threadIdx.x; q < NG; q += 256) __hip_atomic_store(gbar + q * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
  const hipError_t e = wide_regs ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, decoder_mega_kernel<4>, 256, (size_t)lds_bytes)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, decoder_mega_kernel<6>, 256, (size_t)lds_bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
  return nb;
}

void launch_decoder_mega(const MegaLaunch& m, hipStream_t st) {
  if (m.wide_regs)
    hipLaunchKernelGGL(decoder_mega_kernel<4>, dim3(m.groups * m.group_size), dim3(256), m.lds_bytes, st, m.prog, m.nops, m.njobs, m.group_size, m.slots, m.pos, m.n, m.T,
                       m.gbar, m.bar, m.bar_base, m.dbg, m.guard);
  else
    hipLaunchKernelGGL(decoder_mega_kernel<6>, dim3(m.groups * m.group_size), dim3(256), m.lds_bytes, st, m.prog, m.nops, m.njobs, m.group_size, m.slots, m.pos, m.n, m.T,
                       m.gbar, m.bar, m.bar_base, m.dbg, m.guard);
}

}  // namespace cnk
