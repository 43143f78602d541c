// Row-wise kernels of the streaming path: LayerNorm over channels, embedding gather, Emformer
// chunk attention with K/V ring update, prosody cross-attention, uv/f0 head, argmax, and the small
// per-utterance style-pass helpers.  All tensors are channel-last fp32 (see kernels.h: TRef).
// Reductions use 64-lane wavefront shuffles; one wave owns one row.
#include "kernels.h"
#include "rowops.h"

namespace cnk {

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

__device__ __forceinline__ unsigned trow(const TRef& r, int slot, const int* pos, int t) {
  if (r.mode == 0) {
    unsigned p = pos ? (unsigned)pos[slot] : 0u;
    return (p * (unsigned)r.rate + (unsigned)(r.off + t)) & (unsigned)r.lmask;
  }
  return (unsigned)(r.off + t);
}
// (the slot's frame counter already in a register: one dependent load per row instead of one per tensor)
__device__ __forceinline__ float* trowptr_pv(const TRef& r, int i, int slot, unsigned pv, int t) {
  const int s = r.mode == 0 ? slot : i;
  const unsigned row = r.mode == 0 ? ((pv * (unsigned)r.rate + (unsigned)(r.off + t)) & (unsigned)r.lmask) : (unsigned)(r.off + t);
  return r.base + (long long)s * r.slot_stride + (long long)row * r.C;
}
__device__ __forceinline__ float* trowptr(const TRef& r, int i, int slot, const int* pos, int t) {
  int s = r.mode == 0 ? slot : i;
  return r.base + (long long)s * r.slot_stride + (long long)trow(r, s, pos, t) * r.C;
}

// ------------------------------------------------------------------------------------ LayerNorm
__global__ __launch_bounds__(256) void layernorm_kernel(const LNArgs a) { ro::layernorm_tile<false>(a, blockIdx.x); }
void launch_layernorm(const LNArgs& a, hipStream_t st) {
  int rows = a.n * a.T;
  if (rows <= 0) return;
  hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, a);
}

// ------------------------------------------------------------------------------------ copies
__global__ __launch_bounds__(256) void copy_rows_kernel(const CopyArgs a) {
  const int m = blockIdx.x;
  const int i = m / a.T, t = m - i * a.T;
  const int slot = a.slots ? a.slots[i] : i;
  float* y = trowptr(a.y, i, slot, a.pos, t);
  const bool live = !(a.lens && t >= a.lens[i]);
  const float* x = trowptr(a.x, i, slot, a.pos, t);
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) y[c] = live ? x[c] : 0.f;
}
void launch_copy_rows(const CopyArgs& a, hipStream_t st) {
  int rows = a.n * a.T;
  if (rows <= 0) return;
  int bs = a.C >= 256 ? 256 : (a.C > 64 ? 128 : 64);
  hipLaunchKernelGGL(copy_rows_kernel, dim3(rows), dim3(bs), 0, st, a);
}

__global__ __launch_bounds__(256) void embed_kernel(const EmbedArgs a) { ro::embed_tile<false>(a, blockIdx.x); }
void launch_embed(const EmbedArgs& a, hipStream_t st) {
  int rows = a.n * a.T;
  if (rows <= 0) return;
  hipLaunchKernelGGL(embed_kernel, dim3(rows), dim3(a.C >= 256 ? 256 : 64), 0, st, a);
}

// ------------------------------------------------------------------------------------ Emformer attention
// one block per slot, one wave per head.  The slot's queries and this step's keys/values are staged into LDS with
// one batch of loads; lanes own keys (rc | cached left context | utterance) and keep their key/value head-slice in
// registers for all query tokens; softmax and the weighted value sum are 64-lane shuffle reductions.
constexpr int EMF_MAX_DH = 16;
constexpr int EMF_MAX_KPL = 2;   // keys per lane: up to 128 keys
__global__ __launch_bounds__(1024) void emf_attn_kernel(const EmfAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float esm[];
  const int i = blockIdx.x;
  const int slot = a.slots[i];
  const int Q = a.R + a.U;
  const int Qq = Q + (a.M > 0 ? 1 : 0);       // + the summary query
  const int KVR = a.M + Q;                    // rows of this step's key/value projections
  const int dh = a.D / a.H;
  const int past = a.past[slot];
  const int Lc = past < a.LC ? past : a.LC;
  int nm = 0;
  if (a.M > 0) { const int nseg = (past + a.seg - 1) / a.seg; nm = nseg < a.M ? nseg : a.M; }
  const int nk = nm + a.R + Lc + a.U;
  const int h = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* sq = esm;                    // [Qq][D]
  float* skv = esm + Qq * a.D;        // [KVR][2D]
  float* so = skv + KVR * 2 * a.D;    // [Qq][D]
  float* sprob = so + Qq * a.D;       // [H][Qq][KPL*64] normalised attention weights
  float* sval = sprob + a.H * Qq * EMF_MAX_KPL * 64;   // [H][KPL*64][MAX_DH] values of each head
  const float* kvb = a.kv + (long long)i * KVR * 2 * a.D;
  const float* qb = a.q + (long long)i * Qq * a.D;
  for (int e = threadIdx.x; e < Qq * a.D; e += blockDim.x) sq[e] = qb[e] * a.scaling;
  for (int e = threadIdx.x; e < KVR * 2 * a.D; e += blockDim.x) skv[e] = kvb[e];
  const float* kr = a.kring + (long long)slot * a.ring_slot_stride;
  const float* vr = a.vring + (long long)slot * a.ring_slot_stride;
  float kreg[EMF_MAX_KPL][EMF_MAX_DH], vreg[EMF_MAX_KPL][EMF_MAX_DH];
  const int c0 = nm + a.R;            // first cached left-context key
  if (h < a.H) {     // cached left-context keys come straight from the rings (issued before the barrier)
#pragma unroll
    for (int s = 0; s < EMF_MAX_KPL; ++s) {
      const int kk = lane + 64 * s;
      const bool cached = kk >= c0 && kk < c0 + Lc;
      const unsigned r = (unsigned)(past - Lc + (kk - c0)) & (unsigned)a.lmask;
      const float* kp = kr + (long long)(cached ? r : 0) * a.D + h * dh;
      const float* vp = vr + (long long)(cached ? r : 0) * a.D + h * dh;
#pragma unroll
      for (int d = 0; d < EMF_MAX_DH; ++d) {
        kreg[s][d] = (cached && d < dh) ? kp[d] : 0.f;
        vreg[s][d] = (cached && d < dh) ? vp[d] : 0.f;
      }
    }
  }
  __syncthreads();
  if (h < a.H) {
#pragma unroll
    for (int s = 0; s < EMF_MAX_KPL; ++s) {   // memory / right-context / utterance keys of this step from LDS
      const int kk = lane + 64 * s;
      int tok = -1;                            // row of skv
      if (kk < nm) tok = (a.M - nm) + kk;
      else if (kk < c0) tok = a.M + (kk - nm);
      else if (kk >= c0 + Lc && kk < nk) tok = a.M + a.R + (kk - c0 - Lc);
      if (tok >= 0) {
#pragma unroll
        for (int d = 0; d < EMF_MAX_DH; ++d) if (d < dh) { kreg[s][d] = skv[tok * 2 * a.D + h * dh + d]; vreg[s][d] = skv[tok * 2 * a.D + a.D + h * dh + d]; }
      }
    }
    // scores: lanes own keys; two shuffle reductions per query (max, normaliser); the probabilities go to LDS
    float* sp = sprob + h * Qq * EMF_MAX_KPL * 64;
    for (int qi = 0; qi < Qq; ++qi) {
      const float* qp = sq + qi * a.D + h * dh;
      const int kmin = qi >= Q ? nm : 0;      // the summary query does not attend to the memory columns
      float sc[EMF_MAX_KPL];
      float mx = -INFINITY;
#pragma unroll
      for (int s = 0; s < EMF_MAX_KPL; ++s) {
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < EMF_MAX_DH; ++d) if (d < dh) acc += qp[d] * kreg[s][d];
        const int kk = lane + 64 * s;
        sc[s] = (kk < nk && kk >= kmin) ? acc : -INFINITY;
        mx = fmaxf(mx, sc[s]);
      }
      mx = wave_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int s = 0; s < EMF_MAX_KPL; ++s) { const int kk = lane + 64 * s; sc[s] = (kk < nk && kk >= kmin) ? expf(sc[s] - mx) : 0.f; sum += sc[s]; }
      sum = wave_sum(sum);
      const float inv = 1.0f / sum;
#pragma unroll
      for (int s = 0; s < EMF_MAX_KPL; ++s) sp[(qi * EMF_MAX_KPL + s) * 64 + lane] = sc[s] * inv;
    }
    // this head's values to LDS, then lanes own (query, dim) pairs and sum over the keys (no cross-lane reduction)
    float* sv = sval + h * EMF_MAX_KPL * 64 * EMF_MAX_DH;
#pragma unroll
    for (int s = 0; s < EMF_MAX_KPL; ++s)
#pragma unroll
      for (int d = 0; d < EMF_MAX_DH; ++d) sv[(s * 64 + lane) * EMF_MAX_DH + d] = vreg[s][d];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int pair = lane; pair < Qq * dh; pair += 64) {
      const int qi = pair / dh, d = pair - qi * dh;
      float o = 0.f;
      for (int kk = 0; kk < nk; ++kk) {
        const int s = kk >> 6, l = kk & 63;
        o += sp[(qi * EMF_MAX_KPL + s) * 64 + l] * sv[(s * 64 + l) * EMF_MAX_DH + d];
      }
      so[qi * a.D + h * dh + d] = o;
    }
  }
  __syncthreads();
  float* ob = a.out + (long long)i * Qq * a.D;
  for (int e = threadIdx.x; e < Qq * a.D; e += blockDim.x) ob[e] = so[e];
  // append the U new utterance keys/values (rows past .. past+U-1; not read above)
  float* kw = a.kring + (long long)slot * a.ring_slot_stride;
  float* vw = a.vring + (long long)slot * a.ring_slot_stride;
  for (int e = threadIdx.x; e < a.U * a.D; e += blockDim.x) {
    const int u = e / a.D, c = e - u * a.D;
    const unsigned r = (unsigned)(past + u) & (unsigned)a.lmask;
    kw[(long long)r * a.D + c] = skv[(a.M + a.R + u) * 2 * a.D + c];
    vw[(long long)r * a.D + c] = skv[(a.M + a.R + u) * 2 * a.D + a.D + c];
  }
}
void launch_emf_attn(const EmfAttnArgs& a, hipStream_t st) {
  if (a.n <= 0) return;
  const int Q = a.R + a.U, Qq = Q + (a.M > 0 ? 1 : 0), KVR = a.M + Q;
  const size_t smem = ((size_t)Qq * a.D * 2 + (size_t)KVR * 2 * a.D + (size_t)a.H * Qq * EMF_MAX_KPL * 64 + (size_t)a.H * EMF_MAX_KPL * 64 * EMF_MAX_DH) * sizeof(float);
  hipLaunchKernelGGL(emf_attn_kernel, dim3(a.n), dim3(64 * a.H), smem, st, a);
}

// ------------------------------------------------------------------------------------ Emformer memory bank
__global__ __launch_bounds__(256) void emf_mem_prep_kernel(const EmfMemArgs a) {
  const int i = blockIdx.x, slot = a.slots[i];
  const int Q = a.R + a.U, rows = a.M + Q + 1;
  const int past = a.past[slot];
  const int nseg = (past + a.seg - 1) / a.seg;
  const int nm = nseg < a.M ? nseg : a.M;
  float* ln = a.ln + (long long)i * rows * a.D;
  float* bank = a.bank + (long long)slot * a.MB * a.D;
  for (int c = threadIdx.x; c < a.D; c += blockDim.x) {
    // summary = AvgPool1d(kernel = stride = segment) of the normalised utterance (_EmformerLayer.memory_op), first row
    float s = 0.f;
    for (int u = 0; u < a.U; ++u) s += ln[(long long)(a.M + a.R + u) * a.D + c];
    ln[(long long)(a.M + Q) * a.D + c] = s / (float)a.U;
    // the bank as it was before this step (_unpack_state), right-aligned; the entry of segment j lives in ring row j % MB
    for (int m = 0; m < a.M; ++m) {
      const int j = nseg - (a.M - m);
      ln[(long long)m * a.D + c] = (m >= a.M - nm) ? bank[(long long)(j & (a.MB - 1)) * a.D + c] : 0.f;
    }
    // _pack_state: append this step's memory input (MB >= M + 1: the row written is none of those read above)
    bank[(long long)(nseg & (a.MB - 1)) * a.D + c] = a.mems_in[(long long)i * a.D + c];
  }
}
void launch_emf_mem_prep(const EmfMemArgs& a, hipStream_t st) {
  if (a.n <= 0) return;
  hipLaunchKernelGGL(emf_mem_prep_kernel, dim3(a.n), dim3(a.D >= 256 ? 256 : 128), 0, st, a);
}
__global__ void emf_seg_mean_kernel(const float* x, float* out, int rows, int row0, int U, int D) {
  const int i = blockIdx.x;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float s = 0.f;
    for (int u = 0; u < U; ++u) s += x[((long long)i * rows + row0 + u) * D + c];
    out[(long long)i * D + c] = s / (float)U;
  }
}
void launch_emf_seg_mean(const float* x, float* out, int n, int rows, int row0, int U, int D, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(emf_seg_mean_kernel, dim3(n), dim3(128), 0, st, x, out, rows, row0, U, D);
}
__global__ void emf_mem_out_kernel(const float* x, float* out, int rows, int row, int D, int tanh_on_mem) {
  const int i = blockIdx.x;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    const float v = x[((long long)i * rows + row) * D + c];
    out[(long long)i * D + c] = tanh_on_mem ? tanhf(v) : fminf(fmaxf(v, -10.f), 10.f);
  }
}
void launch_emf_mem_out(const float* x, float* out, int n, int rows, int row, int D, int tanh_on_mem, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(emf_mem_out_kernel, dim3(n), dim3(128), 0, st, x, out, rows, row, D, tanh_on_mem);
}

// ------------------------------------------------------------------------------------ cross attention
// block = one query row (slot, t); wave h = head h.  Scores: lane-per-key dot over dh dims with q
// broadcast from LDS; softmax by wave reductions; output: lane-per-dim sum over keys.
using ro::XA_MAX_S; using ro::XA_MAX_H;
__global__ __launch_bounds__(256) void xattn_kernel(const XAttnArgs a) {
  __shared__ __attribute__((aligned(16))) float xl[ro::XA_LDS_FLOATS];
  ro::xattn_tile<false>(a, blockIdx.x, xl);
}
void launch_xattn(const XAttnArgs& a, hipStream_t st) {
  int rows = a.n * a.T;
  if (rows <= 0) return;
  hipLaunchKernelGGL(xattn_kernel, dim3(rows), dim3(64 * (a.H < 1 ? 1 : a.H)), 0, st, a);
}

// ------------------------------------------------------------------------------------ uv / f0 head
// PitchPredictor tail (nar_tts_modules.py:141-146) + add_orig_pitch (Conan.py:330-340) + denorm_f0 /
// f0_to_coarse (pitch/utils.py:71-82, :17-28) + pitch_embed add (Conan.py:301, :181); fp32 op order kept.
__global__ __launch_bounds__(256) void pitch_head_kernel(const PitchHeadArgs a, float mel_min, float mel_den) { ro::pitch_head_tile<false>(a, mel_min, mel_den, blockIdx.x); }
void launch_pitch_head(const PitchHeadArgs& a, hipStream_t st) {
  int rows = a.n * a.T;
  if (rows <= 0) return;
  const double mn = 1127.0 * log(1.0 + 50.0 / 700.0), mxx = 1127.0 * log(1.0 + 900.0 / 700.0);
  hipLaunchKernelGGL(pitch_head_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, a, (float)mn, (float)(mxx - mn));
}

// ------------------------------------------------------------------------------------ MRF mean + LeakyReLU
__global__ __launch_bounds__(256) void mean_act_kernel(const MeanActArgs a) {
  const int c4n = a.C >> 2;
  const long long total = (long long)a.n * a.T * c4n;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % c4n);
    const long long m = e / c4n;
    const int i = (int)(m / a.T), t = (int)(m - (long long)i * a.T);
    const int slot = a.slots ? a.slots[i] : i;
    const unsigned pv = a.pos ? (unsigned)a.pos[slot] : 0u;
    // (the branches' rows in ONE round trip: loads behind `if (nsrc > 1)` are issued one full wait apart; an absent branch re-reads the first)
    const float* p0 = trowptr_pv(a.x[0], i, slot, pv, t) + c4 * 4;
    const float* p1 = a.nsrc > 1 ? trowptr_pv(a.x[1], i, slot, pv, t) + c4 * 4 : p0;
    const float* p2 = a.nsrc > 2 ? trowptr_pv(a.x[2], i, slot, pv, t) + c4 * 4 : p0;
    float4 v = *reinterpret_cast<const float4*>(p0);
    const float4 v1 = *reinterpret_cast<const float4*>(p1);
    const float4 v2 = *reinterpret_cast<const float4*>(p2);
    if (a.nsrc > 1) {
      v.x += v1.x; v.y += v1.y; v.z += v1.z; v.w += v1.w;
      if (a.nsrc > 2) { v.x += v2.x; v.y += v2.y; v.z += v2.z; v.w += v2.w; }
      const float dn = (float)a.nsrc;      // xs / num_resblocks: true division like the reference
      v.x /= dn; v.y /= dn; v.z /= dn; v.w /= dn;
    }
    v.x = v.x > 0.f ? v.x : v.x * a.slope; v.y = v.y > 0.f ? v.y : v.y * a.slope;
    v.z = v.z > 0.f ? v.z : v.z * a.slope; v.w = v.w > 0.f ? v.w : v.w * a.slope;
    *reinterpret_cast<float4*>(trowptr_pv(a.y, i, slot, pv, t) + c4 * 4) = v;
  }
}
void launch_mean_act(const MeanActArgs& a, hipStream_t st) {
  const long long total = (long long)a.n * a.T * (a.C >> 2);
  if (total <= 0) return;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(mean_act_kernel, dim3((int)blocks), dim3(256), 0, st, a);
}

// ------------------------------------------------------------------------------------ conv_post + tanh
constexpr int CP_TILE = 256;
// One thread per output sample: the tile's CP_TILE + k - 1 input rows pass through LDS (rows padded to C + 4 floats: 16-byte
// fragment reads, conflict-free for consecutive rows), the weights sit behind them and are read as broadcasts.  The window is
// fetched eight 16-byte loads per thread at a time - a load per loop iteration made this kernel a chain of nine L2 round trips
// (17.5 us for 10 MB at 64 streams).
__global__ __launch_bounds__(256) void conv_post_kernel(const ConvPostArgs a) {
  extern __shared__ __attribute__((aligned(16))) float cps[];     // [(CP_TILE + k - 1)][C + 4] then w[k][C]
  const int tiles = (a.T + CP_TILE - 1) / CP_TILE;
  const int i = blockIdx.x / tiles, tile = blockIdx.x - i * tiles;
  const int slot = a.slots ? a.slots[i] : i;
  const int t0 = tile * CP_TILE;
  const int rows = CP_TILE + a.k - 1;
  const int ld = a.C + 4;
  float* sw = cps + rows * ld;
  for (int e = threadIdx.x; e < a.k * a.C; e += blockDim.x) sw[e] = a.w[e];
  const int c4n = a.C >> 2;
  // x[] are raw branch outputs (also for a single-branch vocoder: the LeakyReLU in front of conv_post is then applied
  // here) exactly when the caller passes the activated-mean ring
  const bool form = a.xmean.base != nullptr;
  const int total = rows * c4n;
  const unsigned pv = a.pos ? (unsigned)a.pos[slot] : 0u;
  const int nsrc = a.nsrc;
  const float slope = a.slope;
  // (9 units per thread: the 262 x 8 units of the shipped shape - 32 channels, 7 taps - in one pass; every load unconditional with a
  // clamped address: behind `if (t < T) .. if (nsrc > 1)` hipcc waits for each load before it issues the next - two dozen round
  // trips in a kernel with a microsecond of arithmetic)
  constexpr int U = 9;
  for (int e0 = 0; e0 < total; e0 += 256 * U) {
    float4 v0[U], v1[U], v2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + (int)threadIdx.x + 256 * u;
      const int ec = e < total ? e : 0;
      const int r = ec / c4n, c4 = ec - r * c4n;
      const int t = t0 + r - (a.k - 1);            // may be negative: ring history (zeros before stream start)
      const int tc = t < a.T ? t : a.T - 1;          // (rows past the step: read the last row, dropped below)
      const bool hist = form && tc < 0;              // earlier steps: the activated mean as stored
      const float* p0 = (hist ? trowptr_pv(a.xmean, i, slot, pv, tc) : trowptr_pv(a.x[0], i, slot, pv, tc)) + c4 * 4;
      const float* p1 = (form && !hist && nsrc > 1) ? trowptr_pv(a.x[1], i, slot, pv, tc) + c4 * 4 : p0;
      const float* p2 = (form && !hist && nsrc > 2) ? trowptr_pv(a.x[2], i, slot, pv, tc) + c4 * 4 : p0;
      v0[u] = *reinterpret_cast<const float4*>(p0);
      v1[u] = *reinterpret_cast<const float4*>(p1);
      v2[u] = *reinterpret_cast<const float4*>(p2);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + (int)threadIdx.x + 256 * u;
      if (e >= total) continue;
      const int r = e / c4n, c4 = e - r * c4n;
      const int t = t0 + r - (a.k - 1);
      float4 v = t < a.T ? v0[u] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (form && t >= 0 && t < a.T) {        // leaky_relu(mean of the branches): mean_act_kernel's arithmetic, operation for operation
        if (nsrc > 1) { v.x += v1[u].x; v.y += v1[u].y; v.z += v1[u].z; v.w += v1[u].w; }
        if (nsrc > 2) { v.x += v2[u].x; v.y += v2[u].y; v.z += v2[u].z; v.w += v2[u].w; }
        if (nsrc > 1) { const float dn = (float)nsrc; v.x /= dn; v.y /= dn; v.z /= dn; v.w /= dn; }
        v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
        v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
        if (r >= a.k - 1) *reinterpret_cast<float4*>(trowptr_pv(a.xmean, i, slot, pv, t) + c4 * 4) = v;   // this tile's own rows -> the mean ring
      }
      *reinterpret_cast<float4*>(cps + r * ld + c4 * 4) = v;
    }
  }
  __syncthreads();
  const int t = t0 + threadIdx.x;
  if (t < a.T) {
    float acc = 0.f;
    for (int j = 0; j < a.k; ++j) {
      const float4* xr = reinterpret_cast<const float4*>(cps + (threadIdx.x + j) * ld);
      const float4* wr = reinterpret_cast<const float4*>(sw + j * a.C);
      // (the products in channel order, as a scalar loop over c adds them; eight 16-byte fragment pairs in flight per round - one
      // pair per iteration made the loop a chain of LDS latencies)
      for (int c0 = 0; c0 < c4n; c0 += 8) {
        float4 x4[8], w4[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int c = c0 + u < c4n ? c0 + u : c0; x4[u] = xr[c]; w4[u] = wr[c]; }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (c0 + u < c4n) { acc += x4[u].x * w4[u].x; acc += x4[u].y * w4[u].y; acc += x4[u].z * w4[u].z; acc += x4[u].w * w4[u].w; }
      }
    }
    acc += a.bias;
    const long long o = (long long)i * a.T + t;
    if (a.pre) a.pre[o] = acc;
    a.wav[o] = tanhf(acc);
  }
  if (a.adv_pos) {
    // every workgroup read its slot's counter before this point; the one that takes the last ticket advances them all
    __syncthreads();
    __shared__ int s_last;
    if (threadIdx.x == 0) {
      // (relaxed: nothing but the ticket travels between the workgroups - each one's counter reads have returned, their values
      // consumed, before its barrier above.  As an acquire-release operation every one of the 320 tickets wrote back and
      // invalidated its XCD's L2, full of the step's freshly written activations: 18 us for a 5 us kernel.)
      const int ticket = __hip_atomic_fetch_add(a.adv_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = ticket == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (s_last) {
      for (int q = threadIdx.x; q < a.n; q += blockDim.x) a.adv_pos[a.slots ? a.slots[q] : q] += a.adv_delta;
      if (threadIdx.x == 0) __hip_atomic_store(a.adv_ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
void launch_conv_post(const ConvPostArgs& a, hipStream_t st) {
  if (a.n * a.T <= 0) return;
  const int tiles = (a.T + CP_TILE - 1) / CP_TILE;
  const size_t smem = ((size_t)(CP_TILE + a.k - 1) * (a.C + 4) + (size_t)a.k * a.C) * sizeof(float);
  hipLaunchKernelGGL(conv_post_kernel, dim3(a.n * tiles), dim3(256), smem, st, a);
}

// ------------------------------------------------------------------------------------ argmax
__global__ __launch_bounds__(256) void argmax_kernel(const ArgmaxArgs a) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= a.rows) return;
  const float* x = a.x + (long long)r * a.C;
  float best = -INFINITY; int bi = 0x7fffffff;
  for (int c = lane; c < a.C; c += 64) { float v = x[c]; if (v > best) { best = v; bi = c; } }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ob = __shfl_xor(best, o, 64); int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) a.idx[r] = bi;
}
void launch_argmax(const ArgmaxArgs& a, hipStream_t st) {
  if (a.rows <= 0) return;
  hipLaunchKernelGGL(argmax_kernel, dim3((a.rows + 3) / 4), dim3(256), 0, st, a);
}

// ------------------------------------------------------------------------------------ small state ops
__global__ void advance_kernel(int* pos, const int* slots, int n, int delta) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pos[slots[i]] += delta;
}
void launch_advance(int* pos, const int* slots, int n, int delta, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(advance_kernel, dim3((n + 63) / 64), dim3(64), 0, st, pos, slots, n, delta);
}
__global__ void fill_int_kernel(int* p, const int* slots, int n, int value) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[slots[i]] = value;
}
void launch_fill_int(int* p, const int* slots, int n, int value, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(fill_int_kernel, dim3((n + 63) / 64), dim3(64), 0, st, p, slots, n, value);
}
__global__ void copy_int_rows_kernel(int* dst, const int* src, int n, int T, int S) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n * T) { int i = e / T, t = e - i * T; dst[e] = src[i * S + t]; }
}
void launch_copy_int_rows(int* dst, const int* src, int n, int T, int S, hipStream_t st) {
  if (n * T <= 0) return;
  hipLaunchKernelGGL(copy_int_rows_kernel, dim3((n * T + 255) / 256), dim3(256), 0, st, dst, src, n, T, S);
}
__global__ void scatter_int_kernel(int* dst, const int* slots, const int* src, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[slots[i]] = src[i];
}
__global__ void profile_mark_kernel() {}
void launch_profile_mark(hipStream_t st) { hipLaunchKernelGGL(profile_mark_kernel, dim3(1), dim3(64), 0, st); }
__global__ void scatter_ids_kernel(int* dst, const int* slots, const int* src, const int* lens, int n, int S_max) {
  const int i = blockIdx.y, s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < S_max) dst[(long long)slots[i] * S_max + s] = s < lens[i] ? src[(long long)i * S_max + s] : -1;
}
void launch_scatter_ids(int* dst, const int* slots, const int* src, const int* lens, int n, int S_max, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(scatter_ids_kernel, dim3((S_max + 63) / 64, n), dim3(64), 0, st, dst, slots, src, lens, n, S_max);
}
__global__ void gather_ids_kernel(int* dst, int* cnt, const int* src, const int* slen, const int* slots, int n, int S_max) {
  const int i = blockIdx.y, s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < S_max) dst[(long long)i * S_max + s] = src[(long long)slots[i] * S_max + s];
  if (cnt && s == 0) cnt[i] = slen[slots[i]];
}
void launch_gather_ids(int* dst, int* cnt, const int* src, const int* slen, const int* slots, int n, int S_max, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(gather_ids_kernel, dim3((S_max + 63) / 64, n), dim3(64), 0, st, dst, cnt, src, slen, slots, n, S_max);
}
__global__ void scatter_rows_kernel(float* dst, const float* src, const int* slots, int C) {
  const int i = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) dst[(long long)slots[i] * C + c] = src[(long long)i * C + c];
}
void launch_scatter_rows(float* dst, const float* src, const int* slots, int n, int C, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(n), dim3(64), 0, st, dst, src, slots, C);
}
void launch_scatter_int(int* dst, const int* slots, const int* src, int n, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(scatter_int_kernel, dim3((n + 63) / 64), dim3(64), 0, st, dst, slots, src, n);
}
__global__ __launch_bounds__(256) void zero_slots_kernel(float* base, long long slot_stride, long long count,
                                                         const int* slots) {
  float4* p = reinterpret_cast<float4*>(base + (long long)slots[blockIdx.y] * slot_stride);
  const long long n4 = count >> 2;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long long)gridDim.x * blockDim.x)
    p[e] = make_float4(0.f, 0.f, 0.f, 0.f);
}
void launch_zero_slots(float* base, long long slot_stride, long long count, const int* slots, int n, hipStream_t st) {
  if (n <= 0 || count <= 0) return;
  long long n4 = count >> 2;
  int gx = (int)((n4 + 255) / 256); if (gx > 256) gx = 256; if (gx < 1) gx = 1;
  hipLaunchKernelGGL(zero_slots_kernel, dim3(gx, n), dim3(256), 0, st, base, slot_stride, count, slots);
}

// ------------------------------------------------------------------------------------ style-pass helpers
__global__ __launch_bounds__(256) void rowmask_kernel(const RowMaskArgs a) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= a.n * a.T) return;
  const int i = m / a.T, t = m - i * a.T;
  float* mo = trowptr(a.m, i, i, nullptr, t);
  if (a.lens && t >= a.lens[i]) { if (lane == 0) *mo = 0.f; return; }
  const float* x = trowptr(a.x, i, i, nullptr, t);
  float r;
  if (a.mode == 1) r = x[0] != 0.f ? 1.f : 0.f;
  else { float s = 0.f; for (int c = lane; c < a.C; c += 64) s += fabsf(x[c]); s = wave_sum(s); r = s > 0.f ? 1.f : 0.f; }
  if (lane == 0) *mo = r;
}
void launch_rowmask(const RowMaskArgs& a, hipStream_t st) {
  int rows = a.n * a.T; if (rows <= 0) return;
  hipLaunchKernelGGL(rowmask_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, a);
}

__global__ __launch_bounds__(128) void wn_gate_kernel(const WNGateArgs a) {
  const int m = blockIdx.x;
  const int i = m / a.T, t = m - i * a.T;
  if (a.lens && t >= a.lens[i]) return;
  const float* x = trowptr(a.xin, i, i, nullptr, t);
  float* y = trowptr(a.acts, i, i, nullptr, t);
  for (int c = threadIdx.x; c < a.H; c += blockDim.x) {
    float tv = tanhf(x[c]);
    float sv = 1.0f / (1.0f + expf(-x[a.H + c]));
    y[c] = tv * sv;
  }
}
void launch_wn_gate(const WNGateArgs& a, hipStream_t st) {
  int rows = a.n * a.T; if (rows <= 0) return;
  hipLaunchKernelGGL(wn_gate_kernel, dim3(rows), dim3(128), 0, st, a);
}

// WN.forward body after res_skip (wavenet.py:78-85): i < last: x = (x + rs[:H]) * m, out += rs[H:];
// last: out += rs; out *= m.
__global__ __launch_bounds__(128) void wn_update_kernel(const WNUpdateArgs a) {
  const int m = blockIdx.x;
  const int i = m / a.T, t = m - i * a.T;
  if (a.lens && t >= a.lens[i]) return;
  const float* rs = trowptr(a.rs, i, i, nullptr, t);
  float* x = trowptr(a.x, i, i, nullptr, t);
  float* o = trowptr(a.out, i, i, nullptr, t);
  const float mk = *trowptr(a.m, i, i, nullptr, t);
  for (int c = threadIdx.x; c < a.H; c += blockDim.x) {
    float prev = a.first ? 0.f : o[c];
    if (!a.last) { x[c] = (x[c] + rs[c]) * mk; o[c] = prev + rs[a.H + c]; }
    else o[c] = (prev + rs[c]) * mk;
  }
}
void launch_wn_update(const WNUpdateArgs& a, hipStream_t st) {
  int rows = a.n * a.T; if (rows <= 0) return;
  hipLaunchKernelGGL(wn_update_kernel, dim3(rows), dim3(128), 0, st, a);
}

// group_hidden_by_segs (utils/nn/seq_utils.py:307-325) with ids = t//group + 1: mean of each group.
__global__ __launch_bounds__(128) void group_pool_kernel(const PoolArgs a) {
  const int S = (a.T + a.group - 1) / a.group;
  const int m = blockIdx.x;
  const int i = m / S, s = m - i * S;
  const int len = a.lens ? a.lens[i] : a.T;
  float* y = trowptr(a.y, i, i, nullptr, s);
  int t0 = s * a.group, t1 = t0 + a.group; if (t1 > len) t1 = len;
  if (t0 >= len) return;
  const float cnt = (float)(t1 - t0);
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
    float acc = 0.f;
    for (int t = t0; t < t1; ++t) acc += trowptr(a.x, i, i, nullptr, t)[c];
    y[c] = acc / fmaxf(cnt, 1.f);
  }
}
void launch_group_pool(const PoolArgs& a, hipStream_t st) {
  int S = (a.T + a.group - 1) / a.group;
  if (a.n * S <= 0) return;
  hipLaunchKernelGGL(group_pool_kernel, dim3(a.n * S), dim3(128), 0, st, a);
}

// VQEmbeddingEMA eval (prosody_util.py:34-46, :88) + positions (seq_utils.py:6-18, transformer.py:66-67):
// kernel 1: per token argmin over codes of (e2[j] + x2) + (-2*dot) ; z = x + (q - x) -> cat[:, :E]
// kernel 2: per stream running count of tokens with z[0] != 0 -> cat[:, E:2E] = postable[pos]
__global__ __launch_bounds__(64) void vq_argmin_kernel(const VQArgs a) {
  const int lane = threadIdx.x;
  const int m = blockIdx.x;
  const int i = m / a.S, s = m - i * a.S;
  const int len = a.lens ? a.lens[i] : a.S;
  if (s >= len) return;
  const float* x = trowptr(a.x, i, i, nullptr, s);
  const float* dots = trowptr(a.dots, i, i, nullptr, s);
  float x2 = 0.f;
  for (int c = lane; c < a.E; c += 64) x2 += x[c] * x[c];
  x2 = wave_sum(x2);
  float best = INFINITY; int bi = 0x7fffffff;
  for (int j = lane; j < a.M; j += 64) {
    float d = (a.e2[j] + x2) + (-2.0f * dots[j]);
    if (d < best) { best = d; bi = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ob = __shfl_xor(best, o, 64); int oi = __shfl_xor(bi, o, 64);
    if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0 && a.ids) a.ids[(long long)i * a.S_max + s] = bi;
  float* cat = trowptr(a.cat, i, i, nullptr, s);
  const float* q = a.emb + (long long)bi * a.E;
  for (int c = lane; c < a.E; c += 64) { float xv = x[c]; cat[c] = xv + (q[c] - xv); }
}
__global__ __launch_bounds__(256) void vq_pos_kernel(const VQArgs a) {
  __shared__ int spos[XA_MAX_S];
  const int i = blockIdx.x;
  const int len = a.lens ? a.lens[i] : a.S;
  if (threadIdx.x == 0) {
    int cnt = 0;
    for (int s = 0; s < len; ++s) { float z0 = trowptr(a.cat, i, i, nullptr, s)[0]; int nz = z0 != 0.f; cnt += nz; spos[s] = nz ? cnt : 0; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < len * a.E; e += blockDim.x) {
    int s = e / a.E, c = e - s * a.E;
    trowptr(a.cat, i, i, nullptr, s)[a.E + c] = a.postable[(long long)spos[s] * a.E + c];
  }
}
void launch_vq(const VQArgs& a, hipStream_t st) {
  if (a.n * a.S <= 0) return;
  hipLaunchKernelGGL(vq_argmin_kernel, dim3(a.n * a.S), dim3(64), 0, st, a);
  hipLaunchKernelGGL(vq_pos_kernel, dim3(a.n), dim3(256), 0, st, a);
}

// key_padding_mask = tokens[:, :, 0] == 0 (Conan.py:249) as an additive -inf mask per slot
__global__ void kmask_kernel(const KMaskArgs a) {
  const int i = blockIdx.x;
  const int slot = a.slots[i];
  const int len = a.lens ? a.lens[i] : a.S;
  for (int s = threadIdx.x; s < a.S_max; s += blockDim.x) {
    float v = 0.f;
    if (s < len) v = trowptr(a.tok, i, i, nullptr, s)[0] == 0.f ? -INFINITY : 0.f;
    a.kmask[(long long)slot * a.kmask_stride + s] = v;
  }
}
void launch_kmask(const KMaskArgs& a, hipStream_t st) {
  if (a.n <= 0) return;
  hipLaunchKernelGGL(kmask_kernel, dim3(a.n), dim3(128), 0, st, a);
}

// temporal_avg_pool (Conan.py:214-219): sum over non-masked frames / count
__global__ __launch_bounds__(256) void masked_mean_kernel(const MeanArgs a) {
  const int i = blockIdx.x;
  const int slot = a.slots[i];
  const int len = a.lens ? a.lens[i] : a.T;
  float cnt = 0.f;
  for (int t = 0; t < len; ++t) cnt += *trowptr(a.m, i, i, nullptr, t) != 0.f ? 1.f : 0.f;
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
    float acc = 0.f;
    for (int t = 0; t < len; ++t) { float mk = *trowptr(a.m, i, i, nullptr, t); if (mk != 0.f) acc += trowptr(a.x, i, i, nullptr, t)[c]; }
    a.out[(long long)slot * a.out_stride + c] = acc / cnt;
  }
}
void launch_masked_mean(const MeanArgs& a, hipStream_t st) {
  if (a.n <= 0) return;
  hipLaunchKernelGGL(masked_mean_kernel, dim3(a.n), dim3(256), 0, st, a);
}

__global__ __launch_bounds__(256) void mul_mask_kernel(const ScaleMaskArgs a) {
  const int m = blockIdx.x;
  const int i = m / a.T, t = m - i * a.T;
  if (a.lens && t >= a.lens[i]) return;
  const float mk = *trowptr(a.m, i, i, nullptr, t);
  float* x = trowptr(a.x, i, i, nullptr, t);
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) x[c] *= mk;
}
void launch_mul_mask(const ScaleMaskArgs& a, hipStream_t st) {
  int rows = a.n * a.T; if (rows <= 0) return;
  hipLaunchKernelGGL(mul_mask_kernel, dim3(rows), dim3(a.C >= 256 ? 256 : 64), 0, st, a);
}

}  // namespace cnk
