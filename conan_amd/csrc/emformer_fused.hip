// emformer_fused: one launch for the whole streaming Emformer step (torchaudio Emformer.infer: all layers, then the
// output projection and arg-max), replacing ~60 latency-bound launches of the generic kernels.
//
// A step touches only Q = R+U (= 6) tokens per stream and streams are independent, so a block owns G = min(16/Q, 16/H)
// streams as one 16-row tile and walks all layers without leaving the chip.  Activations, the per-stream key tables and
// the layer's small parameters (biases, LayerNorm vectors; staged one layer ahead) live in LDS; weights stream from L2
// straight into MFMA B fragments - they are stored fragment-major (ctx.hip: 1 KiB per 16-column x 16-row operand, in
// the order a wave consumes them), so every load is lane-contiguous - and are refilled in place one phase ahead: the
// block runs one wave per SIMD and has the whole 512-entry register file for that.  Barriers are raw s_barrier so
// those loads stay in flight across them, and nothing but prefetches is loaded from global memory inside the layer
// loop (vmcnt retires in order: a late small load would wait for every prefetch issued before it).
// GEMMs use v_mfma_f32_16x16x4_f32 (exact fp32).  Per layer:
//   LN -> [q|k|v] GEMM (k, v land directly in the per-stream key tables [rc | cached left context | utterance] next
//   to the ring rows prefetched at the top of the layer; the U new rows are appended to the rings) -> attention with
//   16-lane rows per (stream, head), K/V rows in registers, DPP row reductions -> out_proj + residual -> LN -> FFN in
//   256-wide hidden chunks (FF1 -> ReLU -> LDS -> FF2 accumulate, K split over the 4 waves) -> + residual -> LN.
// Token order inside a stream is [right context | utterance] like torchaudio's _EmformerLayer.infer.
//
// Cluster mode (EmfFusedArgs::cs = 2, 4, 8): the step is a chain of latency-bound phases on ONE 16-row tile, 69 % of it
// the feed-forward GEMMs, so with few streams most of the chip idles for 245 us.  cs consecutive workgroups (dealt
// round-robin over the XCDs: one per XCD) then own the same streams; each runs every phase redundantly except the
// feed-forward, of which it takes hidden chunks m, m + cs, ...; every CHUNK's partial [16 x D] sum is formed from zero (the four
// waves' K quarters added in wave order), the chunks' partials meet in global memory once per layer (agent-scope write-through
// stores / sc1 loads, a per-member flag word carrying the launch epoch - no fence, so the weights in this XCD's L2 survive; with
// release / acquire fences the step measured 181 us instead of 141) and are summed in CHUNK order by every member: identical bits in
// every member AND for every cluster size (round 5: until then a member accumulated its chunks in registers and the members' sums
// were added in member order - the cluster size was part of the result's bits, so blocking steps had to use the pipelined steps'
// 64-workgroup split).  Member 0 alone writes rings, outputs and
// past lengths.  One stream: 245 -> 121 us; 64 streams as 32 groups x 4: 147 us.  Forward progress does not need all
// workgroups resident at once: workgroups are dispatched in index order on every XCD, so the members of the earliest
// unfinished cluster are always dispatched before any member of a later one.
#include <atomic>
#include "kernels.h"

namespace cnk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int EF_ROWS = 16;
constexpr int EF_HCHUNK = 256;    // FFN hidden chunk
constexpr int EF_CR = 8;          // ring-prefetch units per thread (one K and one V float4 each)
constexpr int EF_PP = 4;          // float4 parameter-prefetch registers per thread (11 D + F <= 4096 floats)
constexpr int EF_MAXJ = 4;        // keys per lane in a 16-lane row -> <= 64 keys

#ifndef EF_NOLOAD
#define EF_NOLOAD 0   // experiment: skip the FFN weight refills (wrong results, isolates the MFMA time)
#endif
#ifdef EF_STAMPS
#define EF_STAMP(i) do { if (tid == 0 && blockIdx.x == 0 && l == 1) a.dbg[i] = __builtin_readcyclecounter(); } while (0)
#define EF_STAMPC(i) do { if (c0 == member * EF_HCHUNK) EF_STAMP(i); } while (0)
#else
#define EF_STAMP(i) do { } while (0)
#define EF_STAMPC(i) do { } while (0)
#endif

__device__ __forceinline__ void ef_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int CTRL>
__device__ __forceinline__ float ef_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int ef_dppi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
// reductions over a 16-lane row: quad xor 1, quad xor 2, row_half_mirror, row_mirror
__device__ __forceinline__ float ef_row_sum(float v) {
  v += ef_dpp<0xB1>(v); v += ef_dpp<0x4E>(v); v += ef_dpp<0x141>(v); v += ef_dpp<0x140>(v);
  return v;
}
__device__ __forceinline__ float ef_row_max(float v) {
  v = fmaxf(v, ef_dpp<0xB1>(v)); v = fmaxf(v, ef_dpp<0x4E>(v)); v = fmaxf(v, ef_dpp<0x141>(v)); v = fmaxf(v, ef_dpp<0x140>(v));
  return v;
}
__device__ __forceinline__ int ef_row_min(int v) {
  v = min(v, ef_dppi<0xB1>(v)); v = min(v, ef_dppi<0x4E>(v)); v = min(v, ef_dppi<0x141>(v)); v = min(v, ef_dppi<0x140>(v));
  return v;
}

// one fragment-major B operand: 1 KiB per fragment, lane-contiguous (scalar base + the lane's 16-byte slot)
__device__ __forceinline__ float4 ef_frag(const float* __restrict__ w, int idx, unsigned lane16) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(w) + (long long)idx * 1024 + lane16);
}
// acc += A[16][acol0 .. acol0+16*KQ) (LDS) x B
template <int KQ>
__device__ __forceinline__ f32x4 ef_mma(f32x4 acc, const float4 (&b)[KQ], const float* A, int lda, int acol0, int lane) {
  const float* ap = A + (lane & 15) * lda + acol0 + (lane >> 4) * 4;
#pragma unroll
  for (int kq = 0; kq < KQ; ++kq) {
    const float4 a = *reinterpret_cast<const float4*>(ap + kq * 16);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[kq].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[kq].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[kq].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[kq].w, acc, 0, 0, 0);
  }
  return acc;
}

// NT independent 16-column tiles sharing the A rows: one LDS fragment read per k-group, MFMAs of different tiles
// interleaved so no instruction waits on the previous one's accumulator
template <int NT, int KQ>
__device__ __forceinline__ void ef_mma_tiles(f32x4 (&acc)[NT], const float4 (&b)[NT][KQ], const float* A, int lda, int acol0, int lane) {
  const float* ap = A + (lane & 15) * lda + acol0 + (lane >> 4) * 4;
#pragma unroll
  for (int kq = 0; kq < KQ; ++kq) {
    const float4 a = *reinterpret_cast<const float4*>(ap + kq * 16);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[t][kq].x, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[t][kq].y, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[t][kq].z, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[t][kq].w, acc[t], 0, 0, 0);
  }
}

// Same, refilling each k-group's B fragments in place as soon as its MFMAs are issued (refill(kq) loads the
// fragments the same tiles need two phases later): the loads are spread through the MFMA stream, so the matrix and
// memory pipes run concurrently, and every load has a full phase of lead time.
template <int NT, int KQ, class F>
__device__ __forceinline__ void ef_mma_tiles_refill(f32x4 (&acc)[NT], float4 (&b)[NT][KQ], const float* A, int lda, int acol0, int lane, F&& refill) {
  const float* ap = A + (lane & 15) * lda + acol0 + (lane >> 4) * 4;
#pragma unroll
  for (int kq = 0; kq < KQ; ++kq) {
    const float4 a = *reinterpret_cast<const float4*>(ap + kq * 16);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[t][kq].x, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[t][kq].y, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[t][kq].z, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[t][kq].w, acc[t], 0, 0, 0);
    refill(kq);
  }
}
// LayerNorm over D = 16*KQD channels: a 16-lane row per activation row, 4 rows per wave
template <int KQD>
__device__ __forceinline__ void ef_layernorm(const float* src, float* dst, int ld, const float* gamma, const float* beta, int tid) {
  const int r = tid >> 4, c0 = tid & 15;
  constexpr float invD = 1.0f / (16 * KQD);
  float v[KQD], s = 0.f;
#pragma unroll
  for (int i = 0; i < KQD; ++i) { v[i] = src[r * ld + c0 + 16 * i]; s += v[i]; }
  const float mean = ef_row_sum(s) * invD;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < KQD; ++i) { v[i] -= mean; q += v[i] * v[i]; }
  const float rstd = 1.0f / sqrtf(ef_row_sum(q) * invD + 1e-5f);
#pragma unroll
  for (int i = 0; i < KQD; ++i) dst[r * ld + c0 + 16 * i] = v[i] * rstd * gamma[c0 + 16 * i] + beta[c0 + 16 * i];
}

// streams per block: 16 GEMM rows and 16 attention rows (one per (stream, head)) bound it
__host__ __device__ inline int ef_streams_per_block(int Q, int H) { const int g1 = EF_ROWS / Q, g2 = 16 / H; return g1 < g2 ? g1 : g2; }

// floats of the multi-purpose region: FFN hidden chunk | FF2 partial sums | logits
__host__ __device__ inline int ef_scratch_floats(int D, int K) {
  int m = EF_ROWS * (EF_HCHUNK + 4);
  if (4 * EF_ROWS * (D + 4) > m) m = 4 * EF_ROWS * (D + 4);
  if (EF_ROWS * (K + 4) > m) m = EF_ROWS * (K + 4);
  return (m + 3) & ~3;
}

}  // namespace

template <int KQD, int DH, bool MEM>
__global__ __launch_bounds__(256) void emformer_fused_kernel(const EmfFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int D = 16 * KQD;
  constexpr int ld = D + 4, ldk = D + 2, ldh = EF_HCHUNK + 4;
  constexpr int TQ = (3 * KQD + 3) / 4;     // [q|k|v] tiles per wave
  constexpr int TO = (KQD + 3) / 4;         // out_proj tiles per wave
  constexpr int T1 = EF_HCHUNK / 64;        // FF1 tiles per wave per chunk
  constexpr int KQ2 = EF_HCHUNK / 4 / 16;   // FF2 k-groups per wave per chunk
  // per-layer small parameters, staged in LDS one layer ahead (EmfLayerW::params)
  constexpr int PB_BQ = 0, PB_BKV = D, PB_BO = 3 * D, PB_B2 = 4 * D, PB_LNIN = 5 * D, PB_LNFF = 7 * D, PB_LNOUT = 9 * D, PB_B1 = 11 * D;
  static_assert(DH % 2 == 0 && DH <= 16, "head dim");
  // Memory bank (torchaudio max_memory_size = M > 0): a stream takes two more rows of the tile - the summary token (mean of
  // the normalised segment; a query only) and the layer's memory INPUT of this step (raw; through the key / value projection
  // only, its K / V rows join the layer's bank for the following steps: the bank holds projected entries, which is the same
  // arithmetic as projecting the raw entries again every step) - and its key table begins with the bank's valid entries.
  // MEM = false compiles the bank out: the register budget of the M = 0 build (the headline configuration) is not shared with it
  const int M = MEM ? a.M : 0;
  const int R = a.R, U = a.U, Q = R + U, QM = MEM && M > 0 ? Q + 2 : Q, H = a.H, G = ef_streams_per_block(QM, H);
  const int nkmax = M + R + a.LC + U;
  const int par = PB_B1 + a.F;           // floats per layer parameter block (multiple of 4)
  float* X = sm;                         // [16][ld] layer input / residual
  float* Y = X + EF_ROWS * ld;           // [16][ld] LN output
  float* Qb = Y + EF_ROWS * ld;          // [16][ld] scaled queries
  float* ATT = Qb + EF_ROWS * ld;        // [16][ld]
  float* R1 = ATT + EF_ROWS * ld;        // [16][ld]
  float* Hb = R1 + EF_ROWS * ld;         // [16][ldh] FFN hidden chunk; also FF2 partial sums and the logits
  float* PB = Hb + ef_scratch_floats(D, a.K);      // [2][par]
  float* KB = PB + 2 * par;              // [G][nkmax][ldk] keys:  rc | cached | utt
  float* VB = KB + G * nkmax * ldk;      // [G][nkmax][ldk] values
  float* MEMB = VB + G * nkmax * ldk;    // [G][D] the layer's memory input (layer 0: mean of the raw segment; then the clamped summary output)
  float* RED = Hb;                       // [4][16][ld]

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const unsigned lane16 = (unsigned)lane * 16;
  // cluster mode: cs consecutive workgroups (one per XCD: workgroups are dealt round-robin) own the same streams
  const int cs = a.cs, csh = 31 - __builtin_clz(cs);
  const int cluster = (int)(blockIdx.x >> csh), member = (int)(blockIdx.x & (cs - 1));
  const int i0 = cluster * G;
  const int ng = (a.n - i0) < G ? (a.n - i0) : G;      // streams in this block
  const int nrows = ng * QM;
  unsigned xtarget = 0;
  if (cs > 1) xtarget = __hip_atomic_load(a.xepoch + cluster, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;

  // ---- everything that depends on slots / past is computed once: no global loads other than the prefetches below
  // happen inside the layer loop (vmcnt retires in order, so a late small load would wait for the prefetches)
  // (a) this lane's GEMM output rows r = (lane>>4)*4 + r4: key-table row offset and ring append offset
  int tab_off[4], ring_off[4], bank_off[4], rtok[4];
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const int r = (lane >> 4) * 4 + r4, g = r / QM, tok = r - g * QM;
    tab_off[r4] = -1; ring_off[r4] = -1; bank_off[r4] = -1; rtok[r4] = g < ng ? tok : QM;
    if (g < ng) {
      const int slot = a.slots[i0 + g], past = a.past[slot], Lc = past < a.LC ? past : a.LC;
      const int nseg = M > 0 ? (past + U - 1) / U : 0, pm = nseg < M ? nseg : M;      // valid bank entries (_unpack_state)
      if (tok < Q) tab_off[r4] = (g * nkmax + pm + (tok < R ? tok : R + Lc + (tok - R))) * ldk;
      if (tok >= R && tok < Q && member == 0) ring_off[r4] = (int)(slot * a.ring_slot_stride) + (int)((unsigned)(past + tok - R) & (unsigned)a.lmask) * D;
      if (MEM && tok == Q + 1 && member == 0) bank_off[r4] = (int)(slot * a.bank_slot_stride) + (nseg & (a.MB - 1)) * D;      // _pack_state: this step's entry
    }
  }
  // (b) attention: a 16-lane row per (stream, head)
  const int pr = tid >> 4, l16 = tid & 15;
  const bool pok = pr < ng * H;
  const int pg = pok ? pr / H : 0, ph = pok ? pr - pg * H : 0;
  int nk = 0, pmq = 0;                   // keys of the (stream, head) row; of which memory entries (the summary query does not see them)
  if (pok) {
    const int past = a.past[a.slots[i0 + pg]];
    const int nseg = M > 0 ? (past + U - 1) / U : 0;
    pmq = nseg < M ? nseg : M;
    nk = pmq + R + (past < a.LC ? past : a.LC) + U;
  }
  // (c) cached-row prefetch plan (shared by K and V): unit e -> (g, j, c4); ring byte offset (-1: none), table offset
  constexpr int f4 = D / 4;
  int soff[EF_CR], doff[EF_CR];
  {
    const int per_g = (a.LC > 0 ? a.LC : 1) * f4;
#pragma unroll
    for (int i = 0; i < EF_CR; ++i) {
      const int e = tid + 256 * i;
      const int g = __umulhi(e, a.magic_per_g), rem = e - g * per_g, j = rem / f4, c4 = rem - j * f4;
      soff[i] = -1; doff[i] = 0;
      if (g < ng) {
        const int slot = a.slots[i0 + g], past = a.past[slot], Lc = past < a.LC ? past : a.LC;
        const int nseg = M > 0 ? (past + U - 1) / U : 0, pm = nseg < M ? nseg : M;
        if (j < Lc) {
          soff[i] = ((int)(slot * a.ring_slot_stride) + (int)((unsigned)(past - Lc + j) & (unsigned)a.lmask) * D + c4 * 4) * 4;
          doff[i] = (g * nkmax + pm + R + j) * ldk + c4 * 4;
        }
      }
    }
  }
  // (d) bank-row prefetch plan: unit tid -> (g, m, c4), one per thread (G * M * D/4 <= 256)
  int msoff = -1, mdoff = 0;
  if (M > 0) {
    const int per_g = M * f4, g = tid / per_g, rem = tid - g * per_g, m = rem / f4, c4 = rem - m * f4;
    if (g < ng) {
      const int slot = a.slots[i0 + g], past = a.past[slot];
      const int nseg = (past + U - 1) / U, pm = nseg < M ? nseg : M;
      if (m < pm) {       // entry of segment nseg - pm + m lives in bank row (nseg - pm + m) % MB
        msoff = ((int)(slot * a.bank_slot_stride) + ((nseg - pm + m) & (a.MB - 1)) * D + c4 * 4) * 4;
        mdoff = (g * nkmax + m) * ldk + c4 * 4;
      }
    }
  }

  // ---- load the chunk (token order [rc | utt]; the chunk is [utt(U) | rc(R)] per stream) and layer 0's parameters
  for (int e = tid; e < EF_ROWS * ld; e += 256) {
    const int r = e / ld, c = e - r * ld, g = r / QM, tok = r - g * QM;
    float v = 0.f;
    if (r < nrows && tok < Q && c < D) v = a.chunk[((long long)(i0 + g) * Q + (tok < R ? U + tok : tok - R)) * D + c];
    X[e] = v; Y[e] = 0.f; Qb[e] = 0.f; ATT[e] = 0.f; R1[e] = 0.f;
  }
  for (int e = tid; e < par / 4; e += 256) reinterpret_cast<float4*>(PB)[e] = reinterpret_cast<const float4*>(a.layers[0].params)[e];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ef_barrier();
  if (M > 0) {      // _EmformerImpl.infer: the first layer's memory input is the mean of the raw segment
    for (int e = tid; e < G * D; e += 256) {
      const int g = e / D, c = e - g * D;
      float sm_ = 0.f;
      for (int u = 0; u < U; ++u) sm_ += X[(g * QM + R + u) * ld + c];
      MEMB[e] = sm_ / (float)U;
    }
    ef_barrier();
  }

  for (int l = 0; l < a.L; ++l) {
    const EmfLayerW& w = a.layers[l];
    const float* pb = PB + (l & 1) * par;
    EF_STAMP(0);
    // ---- prefetch: cached left-context rows of this layer's rings, [q|k|v] weights
    float4 crk[EF_CR], crv[EF_CR];
    {
      const char* kbase = reinterpret_cast<const char*>(a.kring[l]);
      const char* vbase = reinterpret_cast<const char*>(a.vring[l]);
#pragma unroll
      for (int i = 0; i < EF_CR; ++i) {
        crk[i] = make_float4(0.f, 0.f, 0.f, 0.f); crv[i] = crk[i];
        if (soff[i] >= 0) { crk[i] = *reinterpret_cast<const float4*>(kbase + (unsigned)soff[i]); crv[i] = *reinterpret_cast<const float4*>(vbase + (unsigned)soff[i]); }
      }
    }
    float4 cmk = make_float4(0.f, 0.f, 0.f, 0.f), cmv = cmk;         // this thread's bank row unit
    if (msoff >= 0) {
      cmk = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.bank_k[l]) + (unsigned)msoff);
      cmv = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.bank_v[l]) + (unsigned)msoff);
    }
    float4 bqkv[TQ][KQD];
#pragma unroll
    for (int i = 0; i < TQ; ++i) {
      const int t = wave + 4 * i;
      if (t < 3 * KQD) {
#pragma unroll
        for (int kq = 0; kq < KQD; ++kq) bqkv[i][kq] = ef_frag(w.wqkv, t * KQD + kq, lane16);
      } else {
#pragma unroll
        for (int kq = 0; kq < KQD; ++kq) bqkv[i][kq] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    // 1. layer_norm_input
    ef_layernorm<KQD>(X, Y, ld, pb + PB_LNIN, pb + PB_LNIN + D, tid);
    ef_barrier();
    if (M > 0) {      // summary row = mean of the normalised segment (memory_op); memory-input row = the raw memory input
      for (int e = tid; e < G * D; e += 256) {
        const int g = e / D, c = e - g * D;
        float sm_ = 0.f;
        for (int u = 0; u < U; ++u) sm_ += Y[(g * QM + R + u) * ld + c];
        Y[(g * QM + Q) * ld + c] = sm_ / (float)U;
        Y[(g * QM + Q + 1) * ld + c] = MEMB[e];
      }
      ef_barrier();
    }
    EF_STAMP(1);
    // 2. [q | k | v] = Y x [Wq | Wkv] + b : q (scaled) -> Qb, k / v -> key tables (+ ring append for utterance rows)
    {
      f32x4 accq[TQ];
#pragma unroll
      for (int i = 0; i < TQ; ++i) accq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      ef_mma_tiles<TQ, KQD>(accq, bqkv, Y, ld, 0, lane);
#pragma unroll
      for (int i = 0; i < TQ; ++i) {
        const int t = wave + 4 * i;
        if (t < 3 * KQD) {
          const int cl = lane & 15;
          if (t < KQD) {
            const float bias = pb[PB_BQ + t * 16 + cl];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) Qb[((lane >> 4) * 4 + r4) * ld + t * 16 + cl] = (accq[i][r4] + bias) * a.scaling;
          } else {
            const int ckv = (t - KQD) * 16 + cl;          // column in [k | v]
            const float bias = pb[PB_BKV + ckv];
            const bool isv = (t - KQD) >= KQD;            // wave-uniform (D is a multiple of 16)
            const int c = isv ? ckv - D : ckv;
            float* tab = isv ? VB : KB;
            float* ring = isv ? a.vring[l] : a.kring[l];
            float* bank = !MEM ? nullptr : isv ? a.bank_v[l] : a.bank_k[l];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              const float v = accq[i][r4] + bias;
              if (tab_off[r4] >= 0) tab[tab_off[r4] + c] = v;
              if (ring_off[r4] >= 0) ring[ring_off[r4] + c] = v;
              if (MEM && bank_off[r4] >= 0) bank[bank_off[r4] + c] = v;
            }
          }
        }
      }
    }
    // cached rows -> key tables
#pragma unroll
    for (int i = 0; i < EF_CR; ++i)
      if (soff[i] >= 0) {
        float* dk = KB + doff[i];
        float* dv = VB + doff[i];
        *reinterpret_cast<float2*>(dk) = make_float2(crk[i].x, crk[i].y);
        *reinterpret_cast<float2*>(dk + 2) = make_float2(crk[i].z, crk[i].w);
        *reinterpret_cast<float2*>(dv) = make_float2(crv[i].x, crv[i].y);
        *reinterpret_cast<float2*>(dv + 2) = make_float2(crv[i].z, crv[i].w);
      }
    if (msoff >= 0) {      // bank rows -> the head of the key tables
      float* dk = KB + mdoff;
      float* dv = VB + mdoff;
      *reinterpret_cast<float2*>(dk) = make_float2(cmk.x, cmk.y);
      *reinterpret_cast<float2*>(dk + 2) = make_float2(cmk.z, cmk.w);
      *reinterpret_cast<float2*>(dv) = make_float2(cmv.x, cmv.y);
      *reinterpret_cast<float2*>(dv + 2) = make_float2(cmv.z, cmv.w);
    }
    float4 bo[TO][KQD];
#pragma unroll
    for (int i = 0; i < TO; ++i) {
      const int t = wave + 4 * i;
      if (t < KQD) {
#pragma unroll
        for (int kq = 0; kq < KQD; ++kq) bo[i][kq] = ef_frag(w.wo, t * KQD + kq, lane16);
      }
    }
    ef_barrier();
    EF_STAMP(2);
    // 3. attention: a lane owns keys kk = lane16 + 16 j, keeps their K and V rows in registers, and the softmax /
    //    P.V sums are DPP reductions over the 16-lane row (no LDS round trip)
    {
      const float* kb = KB + pg * nkmax * ldk + ph * DH;
      const float* vb = VB + pg * nkmax * ldk + ph * DH;
      const int jmax = (nkmax + 15) >> 4;               // block-uniform
      float2 kreg[EF_MAXJ][DH / 2], vreg[EF_MAXJ][DH / 2];
#pragma unroll
      for (int j = 0; j < EF_MAXJ; ++j) {
        if (j < jmax) {
          const int kk = l16 + 16 * j;
          const bool ok = kk < nk;
#pragma unroll
          for (int d = 0; d < DH / 2; ++d) {
            kreg[j][d] = ok ? *reinterpret_cast<const float2*>(kb + kk * ldk + 2 * d) : make_float2(0.f, 0.f);
            vreg[j][d] = ok ? *reinterpret_cast<const float2*>(vb + kk * ldk + 2 * d) : make_float2(0.f, 0.f);
          }
        }
      }
      const int nq = MEM && M > 0 ? Q + 1 : Q;            // the summary token is the last query
      for (int qi = 0; qi < nq; ++qi) {
        const int klo = (MEM && qi == Q) ? pmq : 0;           // ... and does not see the memory columns (attention_mask[-1, :mems])
        const float* qp = Qb + (pg * QM + qi) * ld + ph * DH;
        float2 q2[DH / 2];
#pragma unroll
        for (int d = 0; d < DH / 2; ++d) q2[d] = *reinterpret_cast<const float2*>(qp + 2 * d);
        float sc[EF_MAXJ], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < EF_MAXJ; ++j) {
          sc[j] = -INFINITY;
          if (j < jmax) {
            float s0 = 0.f;
#pragma unroll
            for (int d = 0; d < DH / 2; ++d) { s0 += q2[d].x * kreg[j][d].x; s0 += q2[d].y * kreg[j][d].y; }
            sc[j] = ((l16 + 16 * j) < nk && (l16 + 16 * j) >= klo) ? s0 : -INFINITY;
            mx = fmaxf(mx, sc[j]);
          }
        }
        mx = ef_row_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < EF_MAXJ; ++j) { sc[j] = (j < jmax && (l16 + 16 * j) < nk && (l16 + 16 * j) >= klo) ? expf(sc[j] - mx) : 0.f; sum += sc[j]; }
        const float inv = 1.0f / ef_row_sum(sum);
        float o[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) o[d] = 0.f;
#pragma unroll
        for (int j = 0; j < EF_MAXJ; ++j)
          if (j < jmax) {
#pragma unroll
            for (int d = 0; d < DH / 2; ++d) { o[2 * d] += sc[j] * vreg[j][d].x; o[2 * d + 1] += sc[j] * vreg[j][d].y; }
          }
        float mine = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) { const float t = ef_row_sum(o[d]); if (l16 == d) mine = t; }
        if (pok && l16 < DH) ATT[(pg * QM + qi) * ld + ph * DH + l16] = mine * inv;
      }
    }
    ef_barrier();
    EF_STAMP(3);
    // 4. out_proj + residual with the un-normalised layer input -> R1
#pragma unroll
    for (int i = 0; i < TO; ++i) {
      const int t = wave + 4 * i;
      if (t < KQD) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = ef_mma<KQD>(acc, bo[i], ATT, ld, 0, lane);
        const int col = t * 16 + (lane & 15);
        const float bias = pb[PB_BO + col];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int r = (lane >> 4) * 4 + r4;
          float v = acc[r4] + bias + X[r * ld + col];
          if (M > 0 && rtok[r4] >= Q) {
            // the summary row's attention output (no residual) is the NEXT layer's memory input, clamped / tanh'd
            // (_EmformerAttention._forward_impl); it and the memory-input row take no part in the rest of the layer
            if (rtok[r4] == Q) { const float o = acc[r4] + bias; MEMB[(r / QM) * D + col] = a.tanh_on_mem ? tanhf(o) : fminf(fmaxf(o, -10.f), 10.f); }
            v = 0.f;
          }
          R1[r * ld + col] = v;
        }
      }
    }
    // 5. pos_ff: LN -> Linear(D, F) -> ReLU -> Linear(F, D); hidden in EF_HCHUNK-wide chunks, weights one phase ahead
    float4 b1[T1][KQD];
#pragma unroll
    for (int t = 0; t < T1; ++t)
#pragma unroll
      for (int kq = 0; kq < KQD; ++kq) b1[t][kq] = ef_frag(w.w1, ((member * (EF_HCHUNK / 64) + wave) * KQD + kq) * T1 + t, lane16);
    ef_barrier();
    EF_STAMP(4);
    ef_layernorm<KQD>(R1, Y, ld, pb + PB_LNFF, pb + PB_LNFF + D, tid);
    ef_barrier();
    EF_STAMP(5);
    f32x4 acc2[KQD];
    float4 b2[KQD][KQ2];
#pragma unroll
    for (int kq = 0; kq < KQ2; ++kq)
#pragma unroll
      for (int t = 0; t < KQD; ++t) b2[t][kq] = ef_frag(w.w2, ((member * (EF_HCHUNK / 64) + wave) * KQ2 + kq) * KQD + t, lane16);
    // (cluster member m runs hidden chunks m, m + cs, ...)
    constexpr int XE = (EF_ROWS * D + 255) / 256;
    const int cstep = cs * EF_HCHUNK;
    float* const xbase = a.xch + ((long long)cluster * 2 + (l & 1)) * EMF_MAX_CHUNKS * (EF_ROWS * D);
    for (int c0 = member * EF_HCHUNK; c0 < a.F; c0 += cstep) {
      const bool more = c0 + cstep < a.F;
      const int un = (c0 + cstep) / 64 + wave;                   // this wave's 64 hidden columns in its next chunk
      EF_STAMPC(15);
      EF_STAMPC(9);
      // FF1: this wave's quarter of the chunk's hidden columns
      {
        f32x4 acc1[T1];
#pragma unroll
        for (int t = 0; t < T1; ++t) acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        ef_mma_tiles_refill<T1, KQD>(acc1, b1, Y, ld, 0, lane, [&](int kq) {
          if (more && !EF_NOLOAD) {
#pragma unroll
            for (int t = 0; t < T1; ++t) b1[t][kq] = ef_frag(w.w1, (un * KQD + kq) * T1 + t, lane16);
          }
        });
#pragma unroll
        for (int t = 0; t < T1; ++t) {
          const int nl = wave * (EF_HCHUNK / 4) + t * 16;
          const float bias = pb[PB_B1 + c0 + nl + (lane & 15)];
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) { const float v = acc1[t][r4] + bias; Hb[((lane >> 4) * 4 + r4) * ldh + nl + (lane & 15)] = v > 0.f ? v : 0.f; }
        }
      }
      EF_STAMPC(10);
      ef_barrier();
      EF_STAMPC(11);
      EF_STAMPC(12);
      // FF2: all D output columns, this wave's quarter of the chunk's K - from zero: a chunk's partial sum does not depend on which
      // member computes it, or on what that member computed before
#pragma unroll
      for (int t = 0; t < KQD; ++t) acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      ef_mma_tiles_refill<KQD, KQ2>(acc2, b2, Hb, ldh, wave * (EF_HCHUNK / 4), lane, [&](int kq) {
        if (more && !EF_NOLOAD) {
#pragma unroll
          for (int t = 0; t < KQD; ++t) b2[t][kq] = ef_frag(w.w2, (un * KQ2 + kq) * KQD + t, lane16);
        }
      });
      EF_STAMPC(13);
      ef_barrier();
      EF_STAMPC(14);
      // the chunk's partial sum: the waves' K quarters through LDS (RED overlays the hidden chunk, which every wave has read), added in
      // wave order; one workgroup per group keeps the running sum over its chunks - all of them, in chunk order - in ATT, cluster
      // members hand every chunk's partial to the exchange buffer
#pragma unroll
      for (int t = 0; t < KQD; ++t)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) RED[(wave * EF_ROWS + (lane >> 4) * 4 + r4) * ld + t * 16 + (lane & 15)] = acc2[t][r4];
      ef_barrier();
      const int ci = c0 / EF_HCHUNK;
#pragma unroll
      for (int i = 0; i < XE; ++i) {
        const int e = tid + 256 * i;
        if (e < EF_ROWS * D) {
          const int r = e / D, c = e - r * D;
          const float v = (RED[(0 * EF_ROWS + r) * ld + c] + RED[(1 * EF_ROWS + r) * ld + c]) + (RED[(2 * EF_ROWS + r) * ld + c] + RED[(3 * EF_ROWS + r) * ld + c]);
          if (cs == 1) ATT[r * ld + c] = ci == 0 ? v : ATT[r * ld + c] + v;
          else __hip_atomic_store(xbase + (long long)ci * (EF_ROWS * D) + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (more) ef_barrier();        // (the next chunk's FF1 writes the hidden chunk over RED)
    }
    EF_STAMP(6);
    // next layer's parameters: in flight during the tail of this layer, parked in the other LDS parameter block
    float4 np[EF_PP];
    auto load_np = [&]() __attribute__((always_inline)) {
      if (l + 1 < a.L) {
        const float4* src = reinterpret_cast<const float4*>(a.layers[l + 1].params);
#pragma unroll
        for (int i = 0; i < EF_PP; ++i) { const int e = tid + 256 * i; np[i] = make_float4(0.f, 0.f, 0.f, 0.f); if (e < par / 4) np[i] = src[e]; }
      }
    };
    if constexpr (!MEM) load_np();      // (the memory-bank build has no registers left for them across the exchange: behind it)
    EF_STAMP(7);
    // + bias + residual -> ATT (reused as the pre-LN buffer), then layer_norm_output -> X
    if (cs == 1) {
      ef_barrier();
      for (int e = tid; e < EF_ROWS * D; e += 256) {
        const int r = e / D, c = e - r * D;
        ATT[r * ld + c] = ATT[r * ld + c] + pb[PB_B2 + c] + R1[r * ld + c];
      }
    } else {
      // Cluster exchange.  The per-XCD L2s are not coherent with each other: partial sums and flags are written
      // through and read with agent-scope (sc1) accesses, which leaves the weights cached in this XCD's L2 alone (an
      // acquire fence would invalidate them every layer).  The buffer of layer l is reused by layer l + 2: a member
      // writes it only after the exchange of layer l + 1, which every member enters after it has read layer l.
      unsigned* const fl = a.xflag + ((long long)cluster * EMF_MAX_LAYERS + l) * EMF_MAX_CLUSTER;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every wave's partial-sum stores are complete ...
      if (a.fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      ef_barrier();
      if (tid == 0) __hip_atomic_store(fl + member, xtarget, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before the flag goes out
      if (tid < cs) {
        SpinGuard sg;
        const unsigned want = xtarget + a.fault;          // (fault != 0: test hook, a value nobody writes)
        while (__hip_atomic_load(fl + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
          __builtin_amdgcn_s_sleep(2);
          if (spin_expired(sg, a.guard, WAIT_EMF_CLUSTER)) break;     // (bounded: kernels.h, SpinGuard)
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (a.fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      ef_barrier();
      // every chunk's partial, eight at a time (unconditional loads: chunks past the last re-read the last and are dropped), summed in
      // chunk order - the order of a single workgroup's running sum
      const int nch = a.F / EF_HCHUNK;
      constexpr int QB = MEM ? 4 : 8;      // (the memory-bank build sits at the register file's edge: four at a time)
      float sum[XE];
      for (int q0 = 0; q0 < nch; q0 += QB) {
        float pv[QB][XE];
#pragma unroll
        for (int m = 0; m < QB; ++m)
#pragma unroll
          for (int i = 0; i < XE; ++i) {
            const int e = tid + 256 * i;
            const int q = q0 + m < nch ? q0 + m : nch - 1;
            pv[m][i] = __hip_atomic_load(xbase + (long long)q * (EF_ROWS * D) + (e < EF_ROWS * D ? e : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#pragma unroll
        for (int m = 0; m < QB; ++m)
#pragma unroll
          for (int i = 0; i < XE; ++i) {
            const float t = (q0 + m == 0) ? pv[m][i] : sum[i] + pv[m][i];
            sum[i] = (q0 + m < nch) ? t : sum[i];
          }
      }
#pragma unroll
      for (int i = 0; i < XE; ++i) {
        const int e = tid + 256 * i;
        if (e < EF_ROWS * D) {
          const int r = e / D, c = e - r * D;
          ATT[r * ld + c] = sum[i] + pb[PB_B2 + c] + R1[r * ld + c];
        }
      }
    }
    if constexpr (MEM) load_np();
    ef_barrier();
    ef_layernorm<KQD>(ATT, X, ld, pb + PB_LNOUT, pb + PB_LNOUT + D, tid);
    if (M > 0) {      // the summary / memory-input rows carry no layer input (their residual is zero): the row's own lanes clear them
      const int r = tid >> 4, c0 = tid & 15;
      if (r - (r / QM) * QM >= Q) {
#pragma unroll
        for (int i = 0; i < KQD; ++i) X[r * ld + c0 + 16 * i] = 0.f;
      }
    }
    if (l + 1 < a.L) {
      float4* dst = reinterpret_cast<float4*>(PB + ((l + 1) & 1) * par);
#pragma unroll
      for (int i = 0; i < EF_PP; ++i) { const int e = tid + 256 * i; if (e < par / 4) dst[e] = np[i]; }
    }
    ef_barrier();
    EF_STAMP(8);
  }

  // ---- outputs: utterance rows (cluster mode: every member holds the same X; member 0 writes)
  if (member != 0) return;
  if (cs > 1 && tid == 0) __hip_atomic_store(a.xepoch + cluster, xtarget, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (a.out)
    for (int e = tid; e < ng * U * D; e += 256) {
      const int g = e / (U * D), rem = e - g * U * D, u = rem / D, c = rem - u * D;
      a.out[((long long)(i0 + g) * U + u) * D + c] = X[(g * QM + R + u) * ld + c];
    }
  if (a.logits || a.codes) {
    float* LG = Hb;                       // [16][K + 4]
    const int ldl = a.K + 4;
    if (a.wp) {
      for (int t = wave; t < (a.K + 15) / 16; t += 4) {
        float4 b[KQD];
#pragma unroll
        for (int kq = 0; kq < KQD; ++kq) b[kq] = ef_frag(a.wp, t * KQD + kq, lane16);
        const int col = t * 16 + (lane & 15);
        const float bias = col < a.K ? a.bp[col] : 0.f;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = ef_mma<KQD>(acc, b, X, ld, 0, lane);
        if (col < a.K) {
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) LG[((lane >> 4) * 4 + r4) * ldl + col] = acc[r4] + bias;
        }
      }
    } else {
      for (int e = tid; e < EF_ROWS * D; e += 256) { const int r = e / D, c = e - r * D; LG[r * ldl + c] = X[r * ld + c]; }
    }
    ef_barrier();
    // a 16-lane row per utterance token: copy out + first-index arg-max
    for (int ru = tid >> 4; ru < G * U; ru += 16) {
      const int g = ru / U, u = ru - g * U;
      const bool ok = g < ng;
      const float* row = LG + (g * QM + R + u) * ldl;
      const long long orow = (long long)(i0 + g) * U + u;
      float best = -INFINITY; int bi = 0x7fffffff;
      for (int c = l16; c < a.K; c += 16) {
        const float v = row[c];
        if (ok && a.logits) a.logits[orow * a.K + c] = v;
        if (v > best) { best = v; bi = c; }
      }
      const float m = ef_row_max(best);
      bi = ef_row_min(best == m ? bi : 0x7fffffff);
      if (ok && l16 == 0 && a.codes) a.codes[orow] = bi;
    }
  }
  ef_barrier();
  if (tid < ng) a.past[a.slots[i0 + tid]] += U;     // past_length += segment (torchaudio _pack_state)
}

size_t emformer_fused_smem(const EmfFusedArgs& a) {
  const int D = a.D, QM = a.R + a.U + (a.M > 0 ? 2 : 0), G = ef_streams_per_block(QM, a.H);
  const int ld = D + 4, ldk = D + 2, nkmax = a.M + a.R + a.LC + a.U;
  return (size_t)(EF_ROWS * ld * 5 + ef_scratch_floats(D, a.K) + 2 * (11 * D + a.F) + 2 * G * nkmax * ldk + G * D) * sizeof(float);
}

bool emformer_fused_supported(const EmfFusedArgs& a) {
  const int Q = a.R + a.U + (a.M > 0 ? 2 : 0);      // rows per stream (with a memory bank: + summary + memory input)
  if (Q < 1 || Q > EF_ROWS || a.H < 1 || a.H > 16 || a.M < 0) return false;
  const int G = ef_streams_per_block(Q, a.H), nkmax = a.M + a.R + a.LC + a.U, dh = a.D / a.H;
  if (a.M > 0 && (G < 1 || G * a.M * (a.D / 4) > 256 || a.MB <= a.M || (a.MB & (a.MB - 1)) || a.U < 1)) return false;
  return ((a.D == 80 && dh == 10) || (a.D == 64 && dh == 8)) && a.D % a.H == 0 && a.F % EF_HCHUNK == 0 && a.F >= EF_HCHUNK &&
         11 * a.D + a.F <= EF_PP * 256 * 4 && nkmax <= 16 * EF_MAXJ && G * a.LC * (a.D / 4) <= EF_CR * 256 && a.L >= 1 && a.L <= EMF_MAX_LAYERS &&
         (a.wp != nullptr || a.K == a.D) && emformer_fused_smem(a) <= 160 * 1024;
}

template <int KQD, int DH, bool MEM>
static void launch_ef(const EmfFusedArgs& a, hipStream_t st) {
  const int G = ef_streams_per_block(a.R + a.U + (a.M > 0 ? 2 : 0), a.H);
  // the attribute is per device (the code object is loaded once per device): remember which devices have it
  static std::atomic<unsigned long long> attr_devs{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
    (void)hipFuncSetAttribute((const void*)emformer_fused_kernel<KQD, DH, MEM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_devs.fetch_or(bit, std::memory_order_release);
  }
  EmfFusedArgs b = a;
  const int chunks = a.F / EF_HCHUNK;
  if (b.cs < 1 || b.cs > EMF_MAX_CLUSTER || (b.cs & (b.cs - 1)) || chunks % b.cs || chunks > EMF_MAX_CHUNKS || !b.xch || !b.xflag || !b.xepoch) b.cs = 1;
  hipLaunchKernelGGL((emformer_fused_kernel<KQD, DH, MEM>), dim3(((a.n + G - 1) / G) * b.cs), dim3(256), emformer_fused_smem(a), st, b);
}

int emformer_fused_streams_per_block(const EmfFusedArgs& a) { return ef_streams_per_block(a.R + a.U + (a.M > 0 ? 2 : 0), a.H); }

// instantiated shapes: (input_dim, head_dim) = (80, 10) is modules/Emformer/emformer.py's only configuration
void launch_emformer_fused(const EmfFusedArgs& a, hipStream_t st) {
  if (a.n <= 0) return;
  if (a.D == 80 && a.D / a.H == 10) a.M > 0 ? launch_ef<5, 10, true>(a, st) : launch_ef<5, 10, false>(a, st);
  else if (a.D == 64 && a.D / a.H == 8) a.M > 0 ? launch_ef<4, 8, true>(a, st) : launch_ef<4, 8, false>(a, st);
}

}  // namespace cnk
