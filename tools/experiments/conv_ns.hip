// conv_ns: causal conv for the wide, short layers of the vocoder (C = 256 at 32 rows per stream: the ResBlock convs of
// the first HiFi-GAN stage, hifigan_causal.py:230-238 - too few rows per stream for the fused ResBlock pass, too many
// taps x channels for conv_mfma's barrier-per-K-step pipeline, which holds them at ~55 % of the f32 MFMA peak).
//
//   y[i][t][co] = epilogue( sum_j sum_ci W[j][ci][co] * f(x[i][t - (k-1-j)*dil][ci]) ),   f = LeakyReLU(in_slope) or identity
//   epilogue(a) = out_act(a + bias) + res;  optional activated twin y2 = LeakyReLU(y2_slope)(y); optional pixel shuffle
//
// Same operand discipline as resblock_fused.hip (exact-f32 v_mfma_f32_16x16x4_f32, weights fragment-major and private
// to a wave, no barrier inside a K loop), different tile:
//   * a workgroup owns 64 output rows x 64 output columns: the rows are 64/SR segments of SR consecutive rows of one
//     stream (SR = 32: two streams of the first stage), each with its (k-1)*dil rows of left context;
//   * K runs in chunks of 64 input channels: the helper waves (4-7) gather chunk c+1 of the window ([segments x (SR +
//     halo)] rows x 64 channels, LeakyReLU applied on the way) into one of two LDS buffers while the matrix waves (0-3,
//     one per SIMD, one 16-column strip each, four row tiles = 16 MFMAs per 16-deep K group and weight fragment) run
//     chunk c; ONE block barrier per chunk;
//   * accumulators leave through an LDS patch: the helper waves add bias / activation / residual and store 16-byte
//     channel-last rows while the next tile is already running;
//   * persistent workgroups, one per CU, over a longest-first tile list with an agent-scope draw counter (the three
//     branches of a launch cost k = 11 : 7 : 3).
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

#include "conv_ns.h"

namespace cnk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(1)))* ns_gcf4;
typedef f32x4 __attribute__((address_space(1)))* ns_gf4;
typedef const int __attribute__((address_space(1)))* ns_gci;
__device__ __forceinline__ float4 ns_gload4(const float* p) { const f32x4 v = *(ns_gcf4)(p); return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void ns_gstore4(float* p, const float4 v) { *(ns_gf4)(p) = (f32x4){v.x, v.y, v.z, v.w}; }

namespace {
#ifndef NS_PAD
#define NS_PAD 8          // LDS row padding (floats)
#endif
template <int NRW> struct NSGeom {
  static constexpr int TM = 16 * NRW;                  // output rows per tile
  static constexpr int WR_MAX = NRW == 4 ? 168 : 88;   // window rows: segments x (SR + halo), e.g. 2 x (32 + 50) / 32 + 50
  static constexpr int BUF = WR_MAX * (64 + NS_PAD);        // floats per chunk buffer
  static constexpr int NIT = (WR_MAX * 16 + 255) / 256;   // 16-byte window loads per helper thread and chunk
  static constexpr int LDS_FLOATS = 2 * BUF + TM * (64 + 4) + 8;
};
constexpr int NS_TN = 64;                    // output columns per tile (4 matrix waves x 16)
constexpr int NS_CH = 64;                    // input channels per K chunk
constexpr int NS_LDX = NS_CH + NS_PAD;            // LDS row stride of a window chunk (conflict-free ds_read_b128)
constexpr int NS_PLD = NS_TN + 4;            // row stride of the accumulator patch
constexpr int NS_MAXSEG = 4;
}  // namespace

#ifndef NS_ABLATE
#define NS_ABLATE 0     // developer builds (tools/ns_bench): bit 0 no weight refills, bit 1 no A fragment re-reads (wrong results)
#endif
#define NS_SEL(br_, f) ((br_) == 0 ? a.p[0].f : ((br_) == 1 ? a.p[1].f : a.p[2].f))

template <int NRW>
__global__ __launch_bounds__(512, 2) void conv_ns_kernel(const NSArgs a) {
  using G = NSGeom<NRW>;
  constexpr int NS_TM = G::TM, NS_WR_MAX = G::WR_MAX, NS_BUF = G::BUF, NS_NIT = G::NIT;
  __shared__ __attribute__((aligned(16))) float lds[G::LDS_FLOATS];
  float* const patch = lds + 2 * NS_BUF;               // [64][NS_PLD] accumulators of the finished tile
  int* const meta = reinterpret_cast<int*>(patch + NS_TM * NS_PLD);   // [0] next tile index, [1] its branch (-1: none), [2] draw generation, [3] its column tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto tile_word = [&](int idx, int w) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(*(ns_gci)(a.tiles + (long long)idx * 4 + w)); };
  const int ntiles = a.ntiles;
  const int T = a.T, SR = a.SR, NSEG = NS_TM / SR, SPS = T / SR;      // segment rows, segments per tile, segments per stream
  const int Cin = a.Cin, KQ = Cin >> 4, NCH = Cin / NS_CH;

  if (wave >= 4) {
    // ============================================================ helper waves: window gather + output
    const int ht = tid - 256;
#ifndef NS_NOPRIO
    __builtin_amdgcn_s_setprio(3);       // latency-critical: never queue behind the partner wave's MFMA stream
#endif
    auto hbar = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    const float islope = a.in_slope;
    const bool act_in = islope != 1.0f;
    const int xmode = a.p[0].x.mode, xrate = a.p[0].x.rate;      // the branches' tensors share their geometry
    // ---- per-tile gather plan: element u of this thread is window row w = idx / 16, channel quad idx % 16
    long long goff[NS_NIT];               // float offset of the element in chunk 0 (-1: zero fill)
    const float* gx = nullptr;
    auto plan = [&](const int p, const int mt) __attribute__((always_inline)) {
      const int k = NS_SEL(p, k), d = NS_SEL(p, dil), halo = (k - 1) * d, L = SR + halo;
      gx = NS_SEL(p, x.base);
      const long long xss = NS_SEL(p, x.slot_stride);
      const int xoff = NS_SEL(p, x.off);
      const unsigned xm = xmode == 0 ? (unsigned)NS_SEL(p, x.lmask) : 0xffffffffu;
      long long sbase[NS_MAXSEG]; unsigned srow[NS_MAXSEG]; int sok[NS_MAXSEG];
      // lane s looks up segment s (slot, then position): two memory round trips for the whole tile, not two per segment
      int my_slot = 0, my_pos = 0;
      {
        const int sb = mt * NSEG + lane, i = sb / SPS;
        const int ii = (lane < NSEG && i < a.n) ? i : 0;
        my_slot = a.slots ? *(ns_gci)(a.slots + ii) : ii;
        my_pos = a.pos ? *(ns_gci)(a.pos + my_slot) : 0;
      }
#pragma unroll
      for (int s = 0; s < NS_MAXSEG; ++s) {
        const int sb = mt * NSEG + s, i = sb / SPS, t0 = (sb - i * SPS) * SR;
        sok[s] = s < NSEG && i < a.n;
        const int ii = sok[s] ? i : 0;
        const int slot = __builtin_amdgcn_readlane(my_slot, s), pos = __builtin_amdgcn_readlane(my_pos, s);
        sbase[s] = (long long)(xmode == 0 ? slot : ii) * xss;
        srow[s] = (xmode == 0 ? (unsigned)pos * (unsigned)xrate : 0u) + (unsigned)(xoff + t0 - halo);
      }
#pragma unroll
      for (int u = 0; u < NS_NIT; ++u) {
        const int idx = ht + 256 * u, w = idx >> 4, c4 = idx & 15;
        int s = 0;
#pragma unroll
        for (int q = 1; q < NS_MAXSEG; ++q) s += (w >= q * L) ? 1 : 0;
        s = s < NSEG ? s : NSEG - 1;
        const int off = w - s * L;
        long long sb_ = sbase[0]; unsigned sr_ = srow[0]; int ok_ = sok[0];
#pragma unroll
        for (int q = 1; q < NS_MAXSEG; ++q) if (s == q) { sb_ = sbase[q]; sr_ = srow[q]; ok_ = sok[q]; }
        goff[u] = (ok_ && w < NSEG * L) ? sb_ + (long long)((sr_ + (unsigned)off) & xm) * Cin + c4 * 4 : -1ll;
      }
    };
    float4 gv[NS_NIT];
    auto gather_issue = [&](const int c) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < NS_NIT; ++u) gv[u] = goff[u] >= 0 ? ns_gload4(gx + goff[u] + c * NS_CH) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto gather_commit = [&](float* buf) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < NS_NIT; ++u) {
        const int idx = ht + 256 * u, w = idx >> 4, c4 = idx & 15;
        float4 q = gv[u];
        if (act_in) { q.x = fmaxf(q.x, q.x * islope); q.y = fmaxf(q.y, q.y * islope); q.z = fmaxf(q.z, q.z * islope); q.w = fmaxf(q.w, q.w * islope); }   // 0 <= slope < 1
        if (w < NS_WR_MAX) *reinterpret_cast<float4*>(buf + w * NS_LDX + c4 * 4) = q;
      }
    };
    // ---- output of a finished tile: thread -> rows (ht >> 4) + 16 u, channel quad ht & 15
    constexpr int NOUT = NS_TM * (NS_TN / 4) / 256;
    static_assert(NS_TM * (NS_TN / 4) % 256 == 0, "output tile / helper threads");
    float4 oacc[NOUT], ores[NOUT], obias;
    float* optr[NOUT];                    // destination of the row's channel quad (nullptr: row past the batch)
    float* oy2 = nullptr; const float* oybase = nullptr;
    const int oc4 = ht & 15;
    auto out_fetch = [&](const int p, const int mt, const int nt) __attribute__((always_inline)) {
      const int ymode = a.p[0].y.mode, yrate = a.p[0].y.rate, rmode = a.p[0].res.mode, rrate = a.p[0].res.rate;
      const int shuf = a.shuffle_r, Cq = a.Cout / shuf;
      const int col = nt * NS_TN + oc4 * 4;
      int jj = 0, ocol = col;
      if (shuf > 1) { jj = col / Cq; ocol = col - jj * Cq; }
      float* const yb = NS_SEL(p, y.base);
      const long long yss = NS_SEL(p, y.slot_stride);
      const int yC = NS_SEL(p, y.C), yoff = NS_SEL(p, y.off);
      const unsigned ym = ymode == 0 ? (unsigned)NS_SEL(p, y.lmask) : 0xffffffffu;
      const bool has_res = NS_SEL(p, has_res) != 0;
      const float* const rb = NS_SEL(p, res.base);
      const long long rss = NS_SEL(p, res.slot_stride);
      const int rC = NS_SEL(p, res.C), roff = NS_SEL(p, res.off);
      const unsigned rm = rmode == 0 ? (unsigned)NS_SEL(p, res.lmask) : 0xffffffffu;
      oy2 = NS_SEL(p, y2_base); oybase = yb;
      const float* bp = NS_SEL(p, bias);
      obias = bp ? ns_gload4(bp + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < NOUT; ++u) {
        const int r = (ht >> 4) + 16 * u;
        oacc[u] = *reinterpret_cast<const float4*>(patch + r * NS_PLD + oc4 * 4);
        const int s = r / SR, sb = mt * NSEG + s, i = sb / SPS, t = (sb - i * SPS) * SR + (r - s * SR);
        optr[u] = nullptr; ores[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < a.n) {
          const int slot = a.slots ? *(ns_gci)(a.slots + i) : i;
          const int pos = a.pos ? *(ns_gci)(a.pos + slot) : 0;
          const unsigned yrow = ((ymode == 0 ? (unsigned)pos * (unsigned)yrate : 0u) + (unsigned)(yoff + t * shuf + jj)) & ym;
          optr[u] = yb + (long long)(ymode == 0 ? slot : i) * yss + (long long)yrow * yC + ocol;
          if (has_res) {
            const unsigned rrow = ((rmode == 0 ? (unsigned)pos * (unsigned)rrate : 0u) + (unsigned)(roff + t)) & rm;
            ores[u] = ns_gload4(rb + (long long)(rmode == 0 ? slot : i) * rss + (long long)rrow * rC + col);
          }
        }
      }
    };
    auto out_store = [&]() __attribute__((always_inline)) {
      const bool lr = a.out_act == ACT_LRELU;
      const float osl = lr ? a.out_slope : 1.0f, y2s = a.y2_slope;
#pragma unroll
      for (int u = 0; u < NOUT; ++u) {
        if (!optr[u]) continue;
        float4 v = make_float4(oacc[u].x + obias.x, oacc[u].y + obias.y, oacc[u].z + obias.z, oacc[u].w + obias.w);
        v.x *= v.x > 0.f ? 1.0f : osl; v.y *= v.y > 0.f ? 1.0f : osl; v.z *= v.z > 0.f ? 1.0f : osl; v.w *= v.w > 0.f ? 1.0f : osl;
        v.x += ores[u].x; v.y += ores[u].y; v.z += ores[u].z; v.w += ores[u].w;
        ns_gstore4(optr[u], v);
        if (oy2) ns_gstore4(oy2 + (optr[u] - oybase), make_float4(v.x > 0.f ? v.x : v.x * y2s, v.y > 0.f ? v.y : v.y * y2s, v.z > 0.f ? v.z : v.z * y2s, v.w > 0.f ? v.w : v.w * y2s));
      }
    };

    int p = tile_word(blockIdx.x, 0), mt = tile_word(blockIdx.x, 1), nt = tile_word(blockIdx.x, 2);
    if (ht == 0) meta[2] = 0;
    plan(p, mt);
    gather_issue(0);
    gather_commit(lds);
    hbar();                                              // bar #0: chunk 0 of the first tile staged
    int g = 0;                                           // chunks handed over so far (buffer parity)
    int gen = 1;
    int pp = -1, pmt = 0, pnt = 0;                       // finished tile whose accumulators are in the patch
    bool stores_pending = false;
    for (;;) {
      int pn = -1, nidx = 0;
      for (int c = 0; c < NCH; ++c) {
        // the matrix waves run chunk c of this tile now
#if !(NS_ABLATE & 4)
        if (c == 0 && pp >= 0) { out_fetch(pp, pmt, pnt); stores_pending = true; pp = -1; }
#endif
        if (c == NCH - 2) {
          // draw the next tile half way through this one (resblock_fused.hip: keeps the longest-first list balanced)
          if (wave == 4) {
            int nv = 0, pv = -1, ntv = 0;
            if (lane == 0) {
              nv = (int)gridDim.x + __hip_atomic_fetch_add(a.sched, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (nv < ntiles) { pv = *(ns_gci)(a.tiles + (long long)nv * 4); ntv = *(ns_gci)(a.tiles + (long long)nv * 4 + 2); }
            }
            nidx = __builtin_amdgcn_readfirstlane(nv); pn = __builtin_amdgcn_readfirstlane(pv);
            if (lane == 0) {
              meta[0] = nidx; meta[1] = pn; meta[3] = ntv;
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              __hip_atomic_store(&meta[2], gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          } else {
            while (__hip_atomic_load(&meta[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != gen) __builtin_amdgcn_s_sleep(8);
            nidx = __builtin_amdgcn_readfirstlane(meta[0]); pn = __builtin_amdgcn_readfirstlane(meta[1]);
          }
          ++gen;
        }
        bool loading = true;
#if NS_ABLATE & 4
        pn = pn;
        if (c + 1 >= NCH && pn >= 0) plan(pn, tile_word(nidx, 1));
        loading = false;
        if (false)
#endif
        if (c + 1 < NCH) gather_issue(c + 1);
        else if (pn >= 0) { plan(pn, tile_word(nidx, 1)); gather_issue(0); }
        else loading = false;
        if (stores_pending && c == (NCH > 2 ? 1 : 0)) { out_store(); stores_pending = false; }
        if (loading) gather_commit(lds + ((g + 1) & 1) * NS_BUF);
        ++g;
        hbar();                                          // chunk g staged; the matrix waves are done with chunk g - 1
      }
      pp = p; pmt = mt; pnt = nt;
      if (pn < 0) break;
      p = pn; mt = tile_word(nidx, 1); nt = tile_word(nidx, 2);
    }
    if (stores_pending) out_store();
    out_fetch(pp, pmt, pnt);                             // (the matrix waves' last barrier was the one above)
    out_store();
    if (ht == 0) {                                       // the last block to leave re-arms the queue for the next launch
      const int dn = __hip_atomic_fetch_add(a.sched + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (dn == (int)gridDim.x - 1) {
        __hip_atomic_store(a.sched, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.sched + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }

  // ============================================================== matrix waves
#ifdef NS_STAMPS
  unsigned long long st_bar = 0, st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime(), st_first = 0; int st_nb = 0, st_tiles = 0;
  auto bar = [&]() __attribute__((always_inline)) {
    const unsigned long long q0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const unsigned long long q1 = __builtin_amdgcn_s_memtime();
    if (st_nb == 0) st_first = q1 - q0; else st_bar += q1 - q0;
    ++st_nb;
  };
#else
  auto bar = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
#endif
  const int lr = lane & 15, lg = lane >> 4;
  int p = tile_word(blockIdx.x, 0);
  int g = 0;                                              // chunk counter (buffer parity), in step with the helpers
  // weight stream of the current tile and the refill stream two tap-chunks ahead of it
  int k = NS_SEL(p, k), d = NS_SEL(p, dil);
  const float* wl = NS_SEL(p, w) + (long long)(tile_word(blockIdx.x, 2) * 4 + wave) * ((long long)(k + 1) * KQ * 256) + lane * 4;
  const float* wr_ = wl; int kr = k, jr = 0, cr = 0;      // refill stream: tile weights, taps, position (tap, chunk)
  float4 bw[8];
  auto refill4 = [&](const int s0) __attribute__((always_inline)) {   // one (chunk, tap) = 4 K groups into slots s0 .. s0+3, then advance
    const float* src = wr_ + (long long)(jr * KQ + cr * 4) * 256;
#pragma unroll
    for (int q = 0; q < 4; ++q) bw[s0 + q] = ns_gload4(src + q * 256);
    if (++jr == kr) { jr = 0; ++cr; }
  };
  refill4(0);
  __builtin_amdgcn_sched_barrier(0);
  refill4(4);
  __builtin_amdgcn_sched_barrier(0);
  int pn = -1;
  while (p >= 0) {
    const int halo = (k - 1) * d, L = SR + halo;
    int wrow[NRW];                                        // window row of (row tile r, lane row) at tap 0
#pragma unroll
    for (int r = 0; r < NRW; ++r) { const int s = (r * 16) / SR; wrow[r] = s * L + (r * 16 - s * SR) + lr; }
    f32x4 acc[NRW];
#pragma unroll
    for (int r = 0; r < NRW; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tstep = d * NS_LDX;
    const float* wl_next = wl; int k_next = k;            // where the refill stream goes behind this tile (dummy: this tile again)
    // (chunk, tap) pairs two at a time: the 8-slot fragment ring is a compile-time register array
    float4 af[NRW];
    auto tapchunk = [&](const int c, const int j, auto s0_tag) __attribute__((always_inline)) {
      constexpr int s0 = decltype(s0_tag)::value;
      const float* const ab = lds + (g & 1) * NS_BUF + j * tstep + 4 * lg;
      if (j == 0) {
        bar();                                            // chunk c staged (helpers); every matrix wave is done with chunk c - 1
        if (c == NCH - 1) {                               // the helpers drew the next tile one chunk ago
          pn = __builtin_amdgcn_readfirstlane(meta[1]);
          if (pn >= 0) {
            const int ntn = __builtin_amdgcn_readfirstlane(meta[3]);      // (from LDS: a global load here would drain the weight ring)
            k_next = NS_SEL(pn, k);
            wl_next = NS_SEL(pn, w) + (long long)(ntn * 4 + wave) * ((long long)(k_next + 1) * KQ * 256) + lane * 4;
          }
        }
      }
      // this pair's slots are refilled group by group with the (chunk, tap) two ahead (8 groups of lead); behind the tile's
      // last pair that is the next tile's first
      if (cr == NCH) { wr_ = wl_next; kr = k_next; jr = 0; cr = 0; }
      const float* const rsrc = wr_ + (long long)(jr * KQ + cr * 4) * 256;
      if (j == 0) {                                       // first fragments of the chunk (later ones are read one group ahead)
#pragma unroll
        for (int r = 0; r < NRW; ++r) af[r] = *reinterpret_cast<const float4*>(ab + wrow[r] * NS_LDX);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // next group's A rows: same tap, next 16 channels; behind the tap's last group the next tap's first (behind a
        // chunk's last tap: rows of the neighbouring LDS region, unused - the next chunk starts with fresh reads)
        const float* nx = q < 3 ? ab + (q + 1) * 16 : ab + tstep;
        const float4 b = bw[s0 + q];
#pragma unroll
        for (int r = 0; r < NRW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r].x, b.x, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NRW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r].y, b.y, acc[r], 0, 0, 0);
        float4 an[NRW];
#pragma unroll
        for (int r = 0; r < NRW; ++r) {
          acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r].z, b.z, acc[r], 0, 0, 0);
#if NS_ABLATE & 2
          an[r] = af[r];
#else
          an[r] = *reinterpret_cast<const float4*>(nx + wrow[r] * NS_LDX);
#endif
        }
#pragma unroll
        for (int r = 0; r < NRW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r].w, b.w, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NRW; ++r) af[r] = an[r];
        // pin the order: two plain passes, one LDS read (next group's fragment) behind each MFMA of the third, the fourth
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NRW, 0);
#pragma unroll
        for (int r = 0; r < NRW; ++r) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#if !(NS_ABLATE & 1)
        bw[s0 + q] = ns_gload4(rsrc + q * 256);
#endif
        __builtin_amdgcn_sched_group_barrier(0x008, NRW, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      if (++jr == kr) { jr = 0; ++cr; }
    };
    const int npairs = NCH * k;                           // even (NCH is)
    int c = 0, j = 0;
    for (int tc = 0; tc < npairs; tc += 2) {
      tapchunk(c, j, std::integral_constant<int, 0>{});
      if (++j == k) { j = 0; ++c; ++g; }
      tapchunk(c, j, std::integral_constant<int, 4>{});
      if (++j == k) { j = 0; ++c; ++g; }
    }
    // accumulators -> patch (lane (lg, lr): rows 4 lg .. 4 lg + 3 of each row tile, column lr of this wave's strip)
#pragma unroll
    for (int r = 0; r < NRW; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) patch[(r * 16 + 4 * lg + e) * NS_PLD + wave * 16 + lr] = acc[r][e];
    if (pn >= 0) { wl = wl_next; k = k_next; d = NS_SEL(pn, dil); }
    p = pn; pn = -1;
#ifdef NS_STAMPS
    ++st_tiles;
#endif
  }
  bar();                                                  // the last tile's accumulators are in the patch
#ifdef NS_STAMPS
  if (a.dbg && tid == 0) {
    unsigned long long* o = a.dbg + blockIdx.x * 6;
    o[0] = st_bar; o[1] = __builtin_amdgcn_s_memtime() - st_t0; o[2] = __builtin_amdgcn_s_memrealtime() - st_r0; o[3] = st_first; o[4] = st_nb; o[5] = st_tiles;
  }
#endif
}

// ------------------------------------------------------------------------------------------------ host side

static int ns_tile_rows(const NSArgs& a) { return a.rows32 ? 32 : 64; }

bool conv_ns_supported(const NSArgs& a) {
  const int TMh = ns_tile_rows(a);
  if (a.Cin % 128 || a.Cout % NS_TN || a.nprob < 1 || a.nprob > 3) return false;
  if ((a.SR != 16 && a.SR != 32 && a.SR != 64) || a.SR > TMh) return false;
  if (a.T % a.SR) return false;
  if (a.shuffle_r < 1 || a.Cout % a.shuffle_r || ((a.Cout / a.shuffle_r) & 3)) return false;
  for (int p = 0; p < a.nprob; ++p) {
    const NSProb& q = a.p[p];
    if (q.k < 3 || q.k > 16) return false;
    if ((TMh / a.SR) * (a.SR + (q.k - 1) * q.dil) > (TMh == 64 ? NSGeom<4>::WR_MAX : NSGeom<2>::WR_MAX)) return false;
    if (q.x.C != a.Cin || (q.y.C & 3) || (q.has_res && (q.res.C & 3))) return false;
    if (q.x.mode != a.p[0].x.mode || q.x.rate != a.p[0].x.rate || q.y.mode != a.p[0].y.mode || q.y.rate != a.p[0].y.rate) return false;
    if (q.has_res && (q.res.mode != a.p[0].res.mode || q.res.rate != a.p[0].res.rate)) return false;
  }
  return true;
}

int conv_ns_segment_rows(int T, int tile_rows) { return (T % 64 == 0 && tile_rows >= 64) ? 64 : (T % 32 == 0 ? 32 : (T % 16 == 0 ? 16 : 0)); }

// Tile list {branch, m-tile, n-tile, 0}, most expensive branch first, n-tile fastest (the four column tiles of a row
// tile share its window in L2); cached per launch shape in device memory (a handful per model and device, never freed).
static const int* ns_tiles(const NSArgs& a, int mtiles, int ntn, int* total_out) {
  struct Key { int v[8]; bool operator<(const Key& o) const { return memcmp(v, o.v, sizeof(v)) < 0; } };
  static std::map<Key, std::pair<const int*, int>> cache;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  (void)hipGetDevice(&dev);
  Key key = {{a.nprob, mtiles, ntn, a.p[0].k, a.nprob > 1 ? a.p[1].k : 0, a.nprob > 2 ? a.p[2].k : 0, dev, a.rows32}};
  auto it = cache.find(key);
  if (it != cache.end()) { *total_out = it->second.second; return it->second.first; }
  const int per = mtiles * ntn, total = a.nprob * per;
  std::vector<int> order(total);
  for (int i = 0; i < total; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return a.p[x / per].k > a.p[y / per].k; });
  std::vector<int> flat((size_t)(total + 1) * 4, -1);
  for (int e = 0; e < total; ++e) {
    const int id = order[e], p = id / per, rem = id - p * per;
    int* q = flat.data() + (size_t)e * 4;
    q[0] = p; q[1] = rem / ntn; q[2] = rem % ntn; q[3] = 0;
  }
  int* dptr = nullptr;
  if (hipMalloc(&dptr, flat.size() * sizeof(int)) != hipSuccess) { *total_out = 0; return nullptr; }
  (void)hipMemcpy(dptr, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice);
  cache[key] = {dptr, total};
  *total_out = total;
  return dptr;
}

bool launch_conv_ns(const NSArgs& ain, int num_cu, hipStream_t st) {
  NSArgs a = ain;
  const int TMh = ns_tile_rows(a);
  if (a.SR == 0) a.SR = conv_ns_segment_rows(a.T, TMh);
  if (!conv_ns_supported(a) || !a.sched) return false;
  const long long M = (long long)a.n * a.T;
  if (M <= 0) return true;
  const int mtiles = (int)((M + TMh - 1) / TMh), ntn = a.Cout / NS_TN;
  a.tiles = ns_tiles(a, mtiles, ntn, &a.ntiles);
  if (!a.tiles) return false;
  const int grid = std::min(a.ntiles, num_cu);
  if (TMh == 64) hipLaunchKernelGGL(conv_ns_kernel<4>, dim3(grid), dim3(512), 0, st, a);
  else hipLaunchKernelGGL(conv_ns_kernel<2>, dim3(grid), dim3(512), 0, st, a);
  return true;
}

}  // namespace cnk
