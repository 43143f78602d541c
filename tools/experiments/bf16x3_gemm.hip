// Developer experiment: what does the ResBlock K loop deliver when an fp32 product is computed as six bf16 limb products
// (x = h + m + l, three bf16 limbs each operand; hh + hm + mh + hl + mm + lh accumulated in f32 by v_mfma_f32_16x16x32_bf16)
// instead of the exact-f32 MFMA?  Same shape as resblock_fused's c1 phase: a window of rows in LDS (A operand, taps are row
// shifts), weights streamed from L2 as fragment-major private streams (B operand), 4 matrix waves per workgroup each owning
// NCW 16-column tiles for NRW 16-row tiles, one workgroup per CU.  Prints useful TFLOP/s (2*M*N*K) and the error of both
// forms against a float64 sum of the same fp32 inputs.
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/bf16x3_gemm.hip -o /tmp/bf16x3 && /tmp/bf16x3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef const f32x4 __attribute__((address_space(1)))* gcf4;

__device__ __forceinline__ f32x4 gld(const void* p) { return *(gcf4)(p); }

// round-to-nearest-even f32 -> bf16 bits (finite inputs)
__host__ __device__ inline u16 bf16_bits(float f) {
  unsigned u; memcpy(&u, &f, 4);
  return (u16)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__host__ __device__ inline float bf16_val(u16 b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
__host__ __device__ inline void split3(float x, u16& h, u16& m, u16& l) {
  h = bf16_bits(x); const float r1 = x - bf16_val(h);
  m = bf16_bits(r1); const float r2 = r1 - bf16_val(m);
  l = bf16_bits(r2);
}

template <int C>
struct Geo {
  static constexpr int LDB = C + 8;          // bf16 elements per LDS row (16 bytes of padding: conflict-free ds_read_b128)
  static constexpr int LDF = C + 4;          // floats per row of the f32 image
  static constexpr int KB = C / 32;          // 32-deep K blocks per tap (bf16 form)
  static constexpr int KQ = C / 16;          // 16-deep K groups per tap (f32 form)
  static constexpr int NCT = C / 16;
};

// ---- bf16x3 -------------------------------------------------------------------------------------------------------
// weights: [tap][kb][ct][limb][lane] x 16 bytes
template <int C, int NRW, int NCW, int RING>
__global__ __launch_bounds__(512) void k_limb(const float* __restrict__ x, const u16* __restrict__ w, float* __restrict__ out, const int k,
                                              const int dil, const int tiles, const int rows) {
  using G = Geo<C>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u16* P = reinterpret_cast<u16*>(smem);             // [3][rows][LDB]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int plane = rows * G::LDB;
  for (int e = tid; e < rows * (C / 4); e += 512) {
    const int r = e / (C / 4), c4 = e - r * (C / 4);
    const f32x4 v = gld(x + ((long long)(blockIdx.x % 7) * 3 + r) * C + c4 * 4);
    u16 h[4], m[4], l[4];
    for (int i = 0; i < 4; ++i) split3(v[i], h[i], m[i], l[i]);
    u16* d = P + r * G::LDB + c4 * 4;
    *reinterpret_cast<uint2*>(d) = make_uint2(h[0] | (unsigned)h[1] << 16, h[2] | (unsigned)h[3] << 16);
    *reinterpret_cast<uint2*>(d + plane) = make_uint2(m[0] | (unsigned)m[1] << 16, m[2] | (unsigned)m[3] << 16);
    *reinterpret_cast<uint2*>(d + 2 * plane) = make_uint2(l[0] | (unsigned)l[1] << 16, l[2] | (unsigned)l[3] << 16);
  }
  __syncthreads();
  if (wave >= 4) return;
  const int ct0 = wave * NCW;
  const long long kb_stride = (long long)G::NCT * 3 * 512;       // u16 per K block
  const u16* wl = w + ((long long)ct0 * 3 * 64 + lane) * 8;
  const u16* abase = P + (lane & 15) * G::LDB + (lane >> 4) * 8;
  const int nkb = k * G::KB;
  f32x4 acc[NRW][NCW];      // accumulates over the tiles (timing runs); the error check launches one tile
#pragma unroll
  for (int r = 0; r < NRW; ++r)
#pragma unroll
    for (int c = 0; c < NCW; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < tiles; ++t) {
    f32x4 bw[RING][NCW][3];
#pragma unroll
    for (int q = 0; q < RING; ++q)
#pragma unroll
      for (int c = 0; c < NCW; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p) bw[q][c][p] = gld(wl + q * kb_stride + (c * 3 + p) * 512);
    f32x4 af[NRW][3];
#pragma unroll
    for (int r = 0; r < NRW; ++r)
#pragma unroll
      for (int p = 0; p < 3; ++p) af[r][p] = *reinterpret_cast<const f32x4*>(abase + p * plane + r * 16 * G::LDB);
    for (int j = 0; j < k; ++j) {
#pragma unroll
      for (int q = 0; q < G::KB; ++q) {
        const int g = j * G::KB + q;
        // products in the order small -> large
        static constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int s = 0; s < 6; ++s)
#pragma unroll
          for (int r = 0; r < NRW; ++r)
#pragma unroll
            for (int c = 0; c < NCW; ++c)
              acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[r][PA[s]]), __builtin_bit_cast(bf16x8, bw[q % RING][c][PB[s]]), acc[r][c], 0, 0, 0);
        // next A fragments: next K block of this tap, or the next tap's first
        const u16* anext = (q + 1 < G::KB) ? abase + (j * dil) * G::LDB + (q + 1) * 32 : abase + ((j + 1 < k ? j + 1 : 0) * dil) * G::LDB;
#pragma unroll
        for (int r = 0; r < NRW; ++r)
#pragma unroll
          for (int p = 0; p < 3; ++p) af[r][p] = *reinterpret_cast<const f32x4*>(anext + p * plane + r * 16 * G::LDB);
        const int gn = g + RING < nkb ? g + RING : g + RING - nkb;      // wraps to the next tile's first blocks
#pragma unroll
        for (int c = 0; c < NCW; ++c)
#pragma unroll
          for (int p = 0; p < 3; ++p) bw[q % RING][c][p] = gld(wl + gn * kb_stride + (c * 3 + p) * 512);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (blockIdx.x == 0) {
#pragma unroll
    for (int r = 0; r < NRW; ++r)
#pragma unroll
      for (int c = 0; c < NCW; ++c)
        for (int i = 0; i < 4; ++i) out[(r * 16 + (lane >> 4) * 4 + i) * C + (ct0 + c) * 16 + (lane & 15)] = acc[r][c][i];
  } else if (acc[0][0][0] == 12345.678f) out[0] = 1.f;
}

// ---- exact f32 MFMA (the production loop's shape) -------------------------------------------------------------------
// weights: [tap][kq][ct][lane] x 16 bytes (4 consecutive k per lane: k = kq*16 + 4*(lane>>4) + e ... one MFMA per e)
template <int C, int NRW, int NCW, int RING>
__global__ __launch_bounds__(512) void k_f32(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ out, const int k,
                                             const int dil, const int tiles, const int rows) {
  using G = Geo<C>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* X = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int e = tid; e < rows * (C / 4); e += 512) {
    const int r = e / (C / 4), c4 = e - r * (C / 4);
    *reinterpret_cast<f32x4*>(X + r * G::LDF + c4 * 4) = gld(x + ((long long)(blockIdx.x % 7) * 3 + r) * C + c4 * 4);
  }
  __syncthreads();
  if (wave >= 4) return;
  const int ct0 = wave * NCW;
  const long long kq_stride = (long long)G::NCT * 256;
  const float* wl = w + ((long long)ct0 * 64 + lane) * 4;
  const float* abase = X + (lane & 15) * G::LDF + (lane >> 4) * 4;
  const int nkq = k * G::KQ;
  f32x4 acc[NRW][NCW];      // accumulates over the tiles (timing runs); the error check launches one tile
#pragma unroll
  for (int r = 0; r < NRW; ++r)
#pragma unroll
    for (int c = 0; c < NCW; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < tiles; ++t) {
    f32x4 bw[RING][NCW];
#pragma unroll
    for (int q = 0; q < RING; ++q)
#pragma unroll
      for (int c = 0; c < NCW; ++c) bw[q][c] = gld(wl + q * kq_stride + c * 256);
    f32x4 af[NRW];
#pragma unroll
    for (int r = 0; r < NRW; ++r) af[r] = *reinterpret_cast<const f32x4*>(abase + r * 16 * G::LDF);
    for (int j = 0; j < k; ++j) {
#pragma unroll
      for (int q = 0; q < G::KQ; ++q) {
        const int g = j * G::KQ + q;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int r = 0; r < NRW; ++r)
#pragma unroll
            for (int c = 0; c < NCW; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r][e], bw[q % RING][c][e], acc[r][c], 0, 0, 0);
        const float* anext = (q + 1 < G::KQ) ? abase + (j * dil) * G::LDF + (q + 1) * 16 : abase + ((j + 1 < k ? j + 1 : 0) * dil) * G::LDF;
#pragma unroll
        for (int r = 0; r < NRW; ++r) af[r] = *reinterpret_cast<const f32x4*>(anext + r * 16 * G::LDF);
        const int gn = g + RING < nkq ? g + RING : g + RING - nkq;
#pragma unroll
        for (int c = 0; c < NCW; ++c) bw[q % RING][c] = gld(wl + gn * kq_stride + c * 256);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (blockIdx.x == 0) {
#pragma unroll
    for (int r = 0; r < NRW; ++r)
#pragma unroll
      for (int c = 0; c < NCW; ++c)
        for (int i = 0; i < 4; ++i) out[(r * 16 + (lane >> 4) * 4 + i) * C + (ct0 + c) * 16 + (lane & 15)] = acc[r][c][i];
  } else if (acc[0][0][0] == 12345.678f) out[0] = 1.f;
}

static float frand() { return (float)((double)rand() / RAND_MAX * 2.0 - 1.0); }

template <int C, int NRW, int NCW, int RINGL, int RINGF>
void run(int k, int dil, int tiles) {
  using G = Geo<C>;
  const int M = 16 * NRW, rows = M + (k - 1) * dil, xrows = rows + 32;
  std::vector<float> x((size_t)xrows * C), w((size_t)k * C * C);      // w[j][ci][co]
  for (auto& v : x) v = frand() * (rand() % 5 == 0 ? 4.f : 1.f);
  for (auto& v : w) v = frand() * 0.05f;
  // pack
  std::vector<u16> wb((size_t)k * G::KB * G::NCT * 3 * 512);
  for (int j = 0; j < k; ++j)
    for (int kb = 0; kb < G::KB; ++kb)
      for (int ct = 0; ct < G::NCT; ++ct)
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 8; ++e) {
            const int ci = kb * 32 + (l >> 4) * 8 + e, co = ct * 16 + (l & 15);
            u16 p[3]; split3(w[((size_t)j * C + ci) * C + co], p[0], p[1], p[2]);
            for (int q = 0; q < 3; ++q) wb[(((((size_t)j * G::KB + kb) * G::NCT + ct) * 3 + q) * 64 + l) * 8 + e] = p[q];
          }
  std::vector<float> wf((size_t)k * G::KQ * G::NCT * 256);
  float *dx, *dwf, *dout; u16* dwb;
  hipMalloc(&dx, x.size() * 4); hipMalloc(&dwf, wf.size() * 4 + 65536); hipMalloc(&dwb, wb.size() * 2 + 65536); hipMalloc(&dout, (size_t)M * C * 4);
  hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dwf, wf.data(), wf.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dwb, wb.data(), wb.size() * 2, hipMemcpyHostToDevice);
  // reference (block 0 reads x rows from 0)
  std::vector<double> ref((size_t)M * C, 0.0);
  for (int m = 0; m < M; ++m)
    for (int j = 0; j < k; ++j)
      for (int ci = 0; ci < C; ++ci) {
        const double a = x[(size_t)(m + j * dil) * C + ci];
        const float* wr = &w[((size_t)j * C + ci) * C];
        for (int co = 0; co < C; ++co) ref[(size_t)m * C + co] += a * wr[co];
      }
  // f32 form: MFMA e of a 16-deep group takes A column kq*16 + 4*(lane>>4) + e of its row; the weight row is the same channel
  for (int j = 0; j < k; ++j)
    for (int kq = 0; kq < G::KQ; ++kq)
      for (int ct = 0; ct < G::NCT; ++ct)
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 4; ++e) {
            const int ci = kq * 16 + (l >> 4) * 4 + e, co = ct * 16 + (l & 15);
            wf[((((size_t)j * G::KQ + kq) * G::NCT + ct) * 64 + l) * 4 + e] = w[((size_t)j * C + ci) * C + co];
          }
  hipMemcpy(dwf, wf.data(), wf.size() * 4, hipMemcpyHostToDevice);
  const size_t lds_l = (size_t)3 * rows * G::LDB * 2, lds_f = (size_t)rows * G::LDF * 4;
  auto kl = k_limb<C, NRW, NCW, RINGL>;
  auto kf = k_f32<C, NRW, NCW, RINGF>;
  hipFuncSetAttribute((const void*)kl, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flop = 2.0 * M * C * (double)C * k * tiles * 256;
  std::vector<float> got((size_t)M * C);
  for (int form = 0; form < 2; ++form) {
    if ((form == 0 ? lds_l : lds_f) > 160 * 1024) { printf("  %s: LDS %zu KB too large\n", form ? "f32" : "bf16x3", (form == 0 ? lds_l : lds_f) >> 10); continue; }
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0, 0);
      if (form == 0) hipLaunchKernelGGL(kl, dim3(256), dim3(512), lds_l, 0, dx, dwb, dout, k, dil, tiles, rows);
      else hipLaunchKernelGGL(kf, dim3(256), dim3(512), lds_f, 0, dx, dwf, dout, k, dil, tiles, rows);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    if (form == 0) hipLaunchKernelGGL(kl, dim3(256), dim3(512), lds_l, 0, dx, dwb, dout, k, dil, 1, rows);
    else hipLaunchKernelGGL(kf, dim3(256), dim3(512), lds_f, 0, dx, dwf, dout, k, dil, 1, rows);
    hipDeviceSynchronize();
    hipError_t err = hipGetLastError();
    hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost);
    double emax = 0, rms = 0, mag = 0;
    for (size_t i = 0; i < got.size(); ++i) { const double d = got[i] - ref[i]; emax = fmax(emax, fabs(d)); rms += d * d; mag += ref[i] * ref[i]; }
    printf("  %-7s C=%d k=%d dil=%d M=%d NCW=%d LDS %3zu KB: %7.3f ms  %7.1f useful TFLOP/s   max err %.3e  rms err / rms %.3e  (%s)\n", form ? "f32" : "bf16x3",
           C, k, dil, M, NCW, (form == 0 ? lds_l : lds_f) >> 10, best, flop / best / 1e9, emax, sqrt(rms / mag), hipGetErrorString(err));
  }
  hipFree(dx); hipFree(dwf); hipFree(dwb); hipFree(dout);
}


// ---- bf16x3 on v_mfma_f32_32x32x16_bf16: 64 x 64 output per workgroup, a 32 x 32 tile per wave (2 x 2 waves) -------------------
// weights: [tap][kstep of 16 channels][32-column tile][limb][lane] x 16 bytes; LDS rows padded to an ODD number of 16-byte slots
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int C>
__global__ __launch_bounds__(512) void k_limb32(const float* __restrict__ x, const u16* __restrict__ w, float* __restrict__ out, const int k,
                                                const int dil, const int tiles, const int rows) {
  constexpr int LDB = C + 8, KS = C / 16, NCT = C / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u16* P = reinterpret_cast<u16*>(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int plane = rows * LDB;
  for (int e = tid; e < rows * (C / 4); e += 512) {
    const int r = e / (C / 4), c4 = e - r * (C / 4);
    const f32x4 v = gld(x + ((long long)(blockIdx.x % 7) * 3 + r) * C + c4 * 4);
    u16 h[4], m[4], l[4];
    for (int i = 0; i < 4; ++i) split3(v[i], h[i], m[i], l[i]);
    u16* d = P + r * LDB + c4 * 4;
    *reinterpret_cast<uint2*>(d) = make_uint2(h[0] | (unsigned)h[1] << 16, h[2] | (unsigned)h[3] << 16);
    *reinterpret_cast<uint2*>(d + plane) = make_uint2(m[0] | (unsigned)m[1] << 16, m[2] | (unsigned)m[3] << 16);
    *reinterpret_cast<uint2*>(d + 2 * plane) = make_uint2(l[0] | (unsigned)l[1] << 16, l[2] | (unsigned)l[3] << 16);
  }
  __syncthreads();
  if (wave >= 4) return;
  const int wr = wave >> 1, wc = wave & 1;                       // 32-row / 32-column tile of this wave inside the 64 x 64 block tile
  const int ct = (blockIdx.x % (NCT / 2)) * 2 + wc;              // (different blocks take different column pairs)
  const long long ks_stride = (long long)NCT * 3 * 512;          // u16 per K step
  const u16* wl = w + ((long long)ct * 3 * 64 + lane) * 8;
  const u16* abase = P + (wr * 32 + (lane & 31)) * LDB + (lane >> 5) * 8;
  const int nks = k * KS;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  constexpr int RING = 4;
  for (int t = 0; t < tiles; ++t) {
    f32x4 bw[RING][3];
#pragma unroll
    for (int q = 0; q < RING; ++q)
#pragma unroll
      for (int p = 0; p < 3; ++p) bw[q][p] = gld(wl + q * ks_stride + p * 512);
    f32x4 af[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) af[p] = *reinterpret_cast<const f32x4*>(abase + p * plane);
    for (int j = 0; j < k; ++j) {
#pragma unroll
      for (int q = 0; q < KS; ++q) {
        const int g = j * KS + q;
        static constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
        const u16* anext = (q + 1 < KS) ? abase + (j * dil) * LDB + (q + 1) * 16 : abase + ((j + 1 < k ? j + 1 : 0) * dil) * LDB;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[PA[s]]), __builtin_bit_cast(bf16x8, bw[q % RING][PB[s]]), acc, 0, 0, 0);
          if (s == 0) af[2] = *reinterpret_cast<const f32x4*>(anext + 2 * plane);
          if (s == 3) af[1] = *reinterpret_cast<const f32x4*>(anext + plane);
          if (s == 5) af[0] = *reinterpret_cast<const f32x4*>(anext);
        }
        const int gn = g + RING < nks ? g + RING : g + RING - nks;
#pragma unroll
        for (int p = 0; p < 3; ++p) bw[q % RING][p] = gld(wl + gn * ks_stride + p * 512);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (blockIdx.x == 0) {
    // C / D of 32x32: lane l: column l % 32, rows 8 * (i / 4) + (l / 32) * 4 + i % 4
    for (int i = 0; i < 16; ++i) out[(wr * 32 + 8 * (i / 4) + (lane >> 5) * 4 + (i & 3)) * C + ct * 32 + (lane & 31)] = acc[i];
  } else if (acc[0] == 12345.678f) out[0] = 1.f;
}

template <int C>
void run32(int k, int dil, int tiles) {
  constexpr int LDB = C + 8, KS = C / 16, NCT = C / 32;
  const int M = 64, rows = M + (k - 1) * dil, xrows = rows + 32;
  std::vector<float> x((size_t)xrows * C), w((size_t)k * C * C);
  for (auto& v : x) v = frand() * (rand() % 5 == 0 ? 4.f : 1.f);
  for (auto& v : w) v = frand() * 0.05f;
  std::vector<u16> wb((size_t)k * KS * NCT * 3 * 512);
  for (int j = 0; j < k; ++j)
    for (int ks = 0; ks < KS; ++ks)
      for (int ct = 0; ct < NCT; ++ct)
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 8; ++e) {
            const int ci = ks * 16 + (l >> 5) * 8 + e, co = ct * 32 + (l & 31);
            u16 p[3]; split3(w[((size_t)j * C + ci) * C + co], p[0], p[1], p[2]);
            for (int q = 0; q < 3; ++q) wb[(((((size_t)j * KS + ks) * NCT + ct) * 3 + q) * 64 + l) * 8 + e] = p[q];
          }
  float *dx, *dout; u16* dwb;
  hipMalloc(&dx, x.size() * 4); hipMalloc(&dwb, wb.size() * 2 + 65536); hipMalloc(&dout, (size_t)M * C * 4);
  hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dwb, wb.data(), wb.size() * 2, hipMemcpyHostToDevice);
  hipMemset(dout, 0, (size_t)M * C * 4);
  const size_t lds = (size_t)3 * rows * LDB * 2;
  auto kl = k_limb32<C>;
  hipFuncSetAttribute((const void*)kl, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flop = 2.0 * M * 64 * (double)C * k * tiles * 256;
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kl, dim3(256), dim3(512), lds, 0, dx, dwb, dout, k, dil, tiles, rows);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  hipLaunchKernelGGL(kl, dim3(256), dim3(512), lds, 0, dx, dwb, dout, k, dil, 1, rows);
  hipDeviceSynchronize();
  std::vector<float> got((size_t)M * C);
  hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost);
  // block 0 computes columns 0 .. 63 (column tiles 0, 1) of rows 0 .. 63
  double emax = 0, rms = 0, mag = 0;
  for (int m = 0; m < M; ++m)
    for (int co = 0; co < 64; ++co) {
      double ref = 0;
      for (int j = 0; j < k; ++j)
        for (int ci = 0; ci < C; ++ci) ref += (double)x[(size_t)(m + j * dil) * C + ci] * w[((size_t)j * C + ci) * C + co];
      const double d = got[(size_t)m * C + co] - ref; emax = fmax(emax, fabs(d)); rms += d * d; mag += ref * ref;
    }
  printf("  bf16x3 32x32x16  C=%d k=%d dil=%d block tile 64 x 64 (32 x 32 per wave) LDS %3zu KB: %7.3f ms  %7.1f useful TFLOP/s   max err %.3e  rms err / rms %.3e  (%s)\n",
         C, k, dil, lds >> 10, best, flop / best / 1e9, emax, sqrt(rms / mag), hipGetErrorString(hipGetLastError()));
  hipFree(dx); hipFree(dwb); hipFree(dout);
}

int main() {
  srand(1);
  printf("C = 128 (stage 1), 4 matrix waves x 2 column tiles\n");
  run<128, 4, 2, 4, 4>(11, 1, 24);
  run<128, 4, 2, 4, 4>(11, 5, 24);
  run<128, 6, 2, 2, 4>(11, 1, 24);
  run<128, 4, 2, 4, 4>(3, 1, 80);
  printf("C = 64 (stage 2), 4 matrix waves x 1 column tile\n");
  run<64, 8, 1, 2, 4>(11, 1, 24);
  run<64, 11, 1, 2, 4>(11, 1, 24);
  printf("C = 256, block tile 64 x 64 on v_mfma_f32_32x32x16_bf16 (a 32 x 32 tile per wave)\n");
  run32<256>(11, 1, 12);
  run32<256>(3, 5, 40);
  printf("C = 256 (stage 0), 4 matrix waves x 4 column tiles\n");
  run<256, 2, 4, 2, 4>(11, 1, 12);
  run<256, 3, 4, 2, 4>(11, 1, 12);
  return 0;
}
