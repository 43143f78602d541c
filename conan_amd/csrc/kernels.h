// Device-side argument structs and launch wrappers of the conan_hip kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace cnk {

// Developer switches read from the ENVIRONMENT exist only in `make DEV=1` builds (CONAN_DEV_SWITCHES): the shipped library reads
// no environment variable - a deployed process's launch plan is a function of the arguments it passes (conan_streams_opts.flags,
// .dev_plan), not of its environment.
inline const char* dev_getenv(const char* name) {
#ifdef CONAN_DEV_SWITCHES
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}
#define HIP_CHECK(x) ::ch::hip_check((x), #x)


// A channel-last fp32 activation tensor [slot][row][C].
//  mode 0 (ring):   row(t) = (pos[slot] * rate + off + t) & lmask, slot = slots[i]
//  mode 1 (linear): row(t) = off + t,                          slot = i        (scratch / caller buffers)
struct TRef {
  float* base;
  long long slot_stride;  // floats between consecutive slots
  int C;                  // floats per row
  int lmask;              // ring length - 1 (mode 0)
  int rate;               // rows per frame (mode 0)
  int off;                // constant row offset
  int mode;
  int pad_;
};

// The device slot table (conan_streams::d_slots) holds the n active slots followed by kSlotTablePad copies of the last one: kernels
// whose tiles take several whole slots (conv_limb's ragged last tile) read entries past n without a clamp.
constexpr int kSlotTablePad = 32;

// decoder_mega's developer stamp buffer (CONAN_MEGA_STAMPS), in 8-byte words: [0, 128) per-operator end stamps, [128, 512) four
// stage stamps per operator, [512, 512 + 4 * kMegaMaxOps) the gather's stamps, then one XCC-mask word per group.
constexpr int kMegaMaxOps = 80;
constexpr int kMegaDbgGroupWords = 512 + 4 * kMegaMaxOps;
constexpr int kMegaDbgWords = kMegaDbgGroupWords + 64;

enum Act { ACT_NONE = 0, ACT_LRELU = 1, ACT_RELU = 2, ACT_GELU = 3, ACT_TANH = 4 };

// Bounded cross-workgroup waits.  Three kernels wait for other workgroups of their own launch (decoder_mega: group / grid
// barriers; emformer_fused: cluster exchange; resblock_pair: partner flags and the tile mailbox).  Forward progress rests on
// dispatch-order arguments; should one ever be violated, a waiter must not spin for ever - a hung GPU takes the
// whole box down.  Every such poll loop therefore carries a budget: after kSpinBudgetTicks of the 100 MHz s_memrealtime clock
// (50 ms; a healthy wait is microseconds) the waiter writes a code into the stream-set's guard block and LEAVES the wait (the
// launch then finishes with meaningless results); other waiters see the code and leave too.  The host reads the code through a
// host-mapped word at the next stream-ordered entry point and returns CONAN_ERR_HIP from then on (conan_streams::check_fault).
//   guard[0]     error code, device memory (agent scope): 0 = healthy
//   guard[2..3]  address of the host-mapped word the code is copied to
// The clock is read only every 256th poll iteration, and not at all by waits that end earlier.
// What counts against the budget is the time the waiter spent POLLING, not wall time: s_memrealtime keeps running while a queue's
// waves are descheduled (context-save preemption when several processes oversubscribe the hardware queues, a debugger halt), and a
// waiter restored together with its partner would otherwise see the whole pause as "spent" before the partner had a chance to
// arrive.  Two consecutive clock reads are 256 polls apart (~0.1-0.5 ms); a gap above kSpinGapTicks (1 ms) between them is taken
// for a deschedule and counts as 1 ms only (never as nothing: polls slowed down by a congested memory system must still run the
// budget down, a wait without an end takes the box with it).
constexpr unsigned long long kSpinBudgetTicks = 5000000ull;
constexpr unsigned long long kSpinGapTicks = 100000ull;
enum WaitCode { WAIT_MEGA_BARRIER = 1, WAIT_EMF_CLUSTER = 2, WAIT_PAIR_FLAG = 3, WAIT_PAIR_MAILBOX = 4, WAIT_RETIRED_5 = 5 /* (voc_chain, rounds 4-5: tools/experiments/voc_chain) */, WAIT_MEGA_ELECT = 6, WAIT_MEGA_DECIDED = 7, WAIT_MEGA_FLAGS = 8 };
struct SpinGuard { unsigned long long last = 0, acc = 0; unsigned it = 0; };
#if defined(__HIPCC__)
// true: give up (budget spent, or another wait of this stream-set already failed)
__device__ __forceinline__ bool spin_expired(SpinGuard& g, unsigned* guard, unsigned code) {
  if ((++g.it & 255u) != 0u || guard == nullptr) return false;
  if (__hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return true;
  const unsigned long long now = __builtin_amdgcn_s_memrealtime();
  if (g.last != 0) { const unsigned long long d = now - g.last; g.acc += d < kSpinGapTicks ? d : kSpinGapTicks; }
  g.last = now;
  if (g.acc < kSpinBudgetTicks) return false;
  __hip_atomic_store(guard, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned* host = *reinterpret_cast<unsigned* const*>(guard + 2);
  if (host) __hip_atomic_store(host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return true;
}
#endif

// Causal / shifted 1-D convolution as an implicit GEMM on the f32 MFMA:
//   y[i][t][co] = epilogue( sum_{j<ktaps} sum_{ci<Cin} W[j][ci][co] * f(x[i][t + j*dil - pad_left][ci]) )
//   f(v)        = in_act(v)   (LeakyReLU only; the hot layers read tensors their producer stored activated)
//   epilogue(a) = ((out_act((a + bias[co]) * out_scale) + bvec[slot][co]) + res[i][t][co]) * m1[i][t] * m2[i][t]
// and, with shuffle_r > 1, the pixel-shuffled store y[i][t*r + co / Cq][co % Cq] (Cq = Cout / r) for
// weights whose output channels were permuted to j-major at pack time.
struct ConvArgs {
  TRef x;
  TRef y;
  TRef res;
  TRef m1, m2;          // per-row masks (C == 1)
  const float* w;       // packed [ktaps][Cin_pad/4][Cout_pad][4]
  const float* bias;    // [Cout_pad] or nullptr
  const float* bvec;    // per-slot broadcast vector [slot][bvec_stride] or nullptr (indexed by slots[i])
  const int* slots;     // [n]
  const int* pos;       // per-slot frame counters (indexed by slot) or nullptr
  const int* lens;      // per-batch valid output rows (or nullptr: all T rows valid)
  long long bvec_stride;
  int has_res, has_m1, has_m2;
  int Cin, Cin_pad, Cout, Cout_pad;
  int ktaps, dil, pad_left;
  int T;                // output rows per slot
  int n;                // slots in this batch
  int in_act;
  float in_slope;
  int out_act;
  float out_scale, out_slope;
  int shuffle_r;
  int Cin_alloc;        // packed rows per tap (Cin rounded up to 128, zero filled): K-steps may over-read safely
  // optional second output: LeakyReLU(y2_slope) of the stored value, written to a tensor with y's geometry (only the
  // base differs).  The consumer then reads pre-activated rows and its K loop needs no input transform.
  float* y2_base;
  float y2_slope;
  // the weights as bf16 limbs for conv_limb.hip (null: not packed): [Cout_pad16/16 column tiles][Cin/32 channel blocks][ktaps]
  // [3 limbs][64 lanes][8], lane = (packed column ct*16 + (lane & 15), channels cb*32 + 8*(lane >> 4) + e)
  const unsigned short* wl;
#ifdef CK_STAMPS
  unsigned long long* dbg;   // developer build only (tools/conv_bench -DCK_STAMPS): s_memtime stamps of block 1
#endif
};

struct ConvGroup {      // up to 3 independent problems in one launch
  ConvArgs p[3];
  int tile_start[4];    // filled by launch_conv: first tile of the q-th scheduled problem; [3] = total tiles
  int tiles_n[3];       // n-tiles of the q-th scheduled problem
  int order[3];         // q-th scheduled problem -> index into p[] (longest K first)
  // inter-block split-K (latency-bound layers with few tiles): K-steps of a tile are divided over `ksplit` blocks;
  // partial tiles go to `slab` ([tile][slice][TM*TN] floats), `counters` ([tile] ints, zero between launches)
  // elects the last-arriving block as the reducer.
  float* slab;
  int* counters;
  // balanced static schedule (filled by launch_conv for persistent launches): block b runs items
  // assign[b * assign_per + i], i = 0 .. until -1.  nullptr = round-robin over the grid.
  const int* assign;
  int assign_per;
  int ksplit;
  // tiles [0, split_from) are whole work items; only the tiles from split_from on are divided into `ksplit` K slices (the
  // tail of a launch whose tile count is not a multiple of the CU count: 320 tiles on 256 CUs = 256 tiles + 64 x 4
  // quarter tiles - every CU gets 1.25 tiles of work instead of some getting two).  0: every tile is split.
  int split_from;
  // 1: the split-K hand-off is additionally bracketed by agent-scope release / acquire fences (CONAN_FENCED=1, a
  // developer cross-check of the fence-free default: write-through stores, ticket, sc1 loads)
  int fenced;
};

// Tile configurations of conv_mfma (block = 256 threads = 4 waves).
enum ConvCfg { CFG_128x64 = 0, CFG_64x64 = 1, CFG_128x32 = 2, CFG_32x64_K2 = 3, CFG_32x32_K4 = 4, CFG_64x32_K2 = 5,
               CFG_64x64_KS64 = 6, CFG_128x32_KS64 = 7, NUM_CFG };
int conv_cfg_ks(int cfg);          // K-step of the configuration
bool conv_cfg_splitk(int cfg);     // the build carries the inter-block split-K hand-off
const char* conv_cfg_name(int cfg);  // the kernel's name as rocprofv3 prints it
int conv_cfg_tm(int cfg);
int conv_cfg_tn(int cfg);
void launch_conv(const ConvGroup& g, int nprob, int cfg, hipStream_t st, int num_cu = 256);

// A convolution with few rows per slot and a long K (ups.0) as a split-K GEMM in the limb arithmetic (conv_tall.hip): tiles of 128 rows
// of any slots x 128 columns, items = (tile, K slice), the slices' partial tiles summed in slice order by the last arrival.
struct ConvTallArgs {
  ConvArgs p;
  int S, nbps;            // K slices per tile, (tap, 32-channel) blocks per slice (even)
  int mtiles, ntiles;
  float* slab;            // [tile][slice][128 x 128] partial tiles
  int* counters;          // [tile][4 matrix waves] tickets, zero before and after every launch
};
bool conv_tall_supported(const ConvArgs& a);
bool conv_tall_plan(const ConvArgs& a, int num_cu, int plan_n, ConvTallArgs* out, long long slab_floats, int max_counters, int max_T = 32);
void launch_conv_tall(const ConvTallArgs& g, hipStream_t st);

// The same convolution with every fp32 product as six bf16 limb products on the bf16 MFMA (conv_limb.hip); covers the
// subset of ConvArgs that conv_limb_supported() accepts.
struct ConvLimbGroup {
  ConvArgs p[3];        // problems with the same n, T and column count (different taps / dilations / tensors)
  int nprob;
  const int* tiles;     // filled by launch_conv_limb: [items] {problem, m tile, n tile, K slice | slices << 8 | split-tile index << 16}
  const int* assign;    // [grid][assign_per] item indices per block, -1 terminated
  int assign_per;
  int wr_max;           // window rows of the largest problem's tile
  // split-K tail (a single problem whose tile count is not a multiple of the CU count - ups.1: 320 tiles on 256 CUs): the last
  // tiles % CUs tiles are cut into K slices over their channel blocks; partial tiles meet in `slab` ([split tile][slice][TM x TN]),
  // one ticket per (split tile, matrix wave) in `counters` (zero before and after every launch); nullptr: no split
  float* slab; int* counters;
  long long slab_floats; int max_counters;
};
bool conv_limb_supported(const ConvArgs& a);
int conv_limb_shape(const ConvArgs* p, int nprob, int num_cu, int plan_n = 0, bool tail_split = false);      // tile shape for these problems, -1: none fits (plan_n: conv_limb.hip)
bool launch_conv_limb(const ConvLimbGroup& g, int shape, int num_cu, hipStream_t st);
int conv_limb_tail_slices(const ConvArgs& a, int shape, int num_cu);      // K slices of the split tail this launch would run with (1: none)
const char* conv_limb_name(int shape);

// One ResBlock1 unit (hifigan_causal.py:230-238) as ONE tile pass (resblock_fused.hip):
//   y = c2(leaky_relu(c1(leaky_relu(x)))) + x,  c1: k taps, dilation dil;  c2: k taps, dilation 1, both causal.
struct RBProb {
  const float* w1; const float* w2;   // fragment-major: [C/16 column tiles][k + 1 taps (last zero)][C/16 K groups][64 lanes][4]
  const float* b1; const float* b2;   // [C]
  const unsigned short* w1l; const unsigned short* w2l;   // the same weights as bf16 limbs (resblock_limb.hip):
                                      // [C/16 column tiles][k + 1 taps (last zero)][C/32 K blocks][3 limbs][64 lanes][8]; null: not packed
  TRef x;                             // raw input (c1 operand after LeakyReLU, residual operand as it is)
  TRef y;                             // raw output
  int k, dil;
};
struct RBArgs {
  RBProb p[3];                        // the branches of a stage at one dilation index (inputs / outputs share their ring geometry's mode and rate)
  const int* slots; const int* pos;
  const int* tiles;                   // filled by launch_resblock_fused: [ntiles] {branch, slot index, first row, 0}, most expensive first
                                      // (merge: (slot, row tile)-major, branches 0 .. nprob-1 adjacent)
  int ntiles;
  int* sched;                         // work-queue state owned by the caller: 2 ints, zero before the first launch (re-armed by the kernel);
                                      // one per stream that may have a fused launch in flight
  int nprob, n, T;                    // branches, slots in this batch, output rows per slot
  int tiles_per_slot;
  int order[3];                       // the branches by descending k (stable): the order of the separate-branch tile list (filled by the launcher)
  float slope;
  // merge = 1 (last dilation of a stage, enough row tiles to fill the chip): a workgroup runs the nprob branches of one
  // (slot, row tile) one after the other and stores only leaky_relu(mean of the branch outputs) to `ymean` - the sum
  // (v0 + v1) + v2, the division and the activation are mean_act_kernel's, operation for operation, so the result does
  // not depend on whether a launch was merged; the branches' own outputs (p[].y) are not written.
  int merge;
  TRef ymean;
  unsigned long long* dbg;            // developer builds only (tools/rb_bench -DRB_ABLATE=4): cycle stamps per block
};
bool resblock_fused_supported(int C, int kmax, int span_max);
int resblock_fused_rows(int C, int T, int n, int ksum, int kmax, int num_cu);   // output rows per tile to launch with
bool launch_resblock_fused(const RBArgs& a, int C, int rows, int num_cu, hipStream_t st);
bool resblock_fused_can_merge(int C, int rows);   // a merged-branch build exists for this geometry
const char* resblock_fused_name(int C, int rows, bool merge = false);
const int* resblock_tiles(const RBArgs& a, int ro, int* total_out);      // the cached device tile list of a launch shape (resblock_fused.hip)
// The same tile pass with every fp32 product as six bf16 limb products on the bf16 MFMA (resblock_limb.hip): same arguments
// (RBProb::w1l / w2l), half as tall tiles.
bool resblock_limb_supported(int C, int kmax, int span_max);      // (and at most 256 slots per launch: launch_resblock_limb refuses more)
constexpr int kResblockLimbMaxSlots = 256;
int resblock_limb_rows(int C, int span_max);
bool launch_resblock_limb(const RBArgs& a, int C, int rows, int num_cu, hipStream_t st);
bool resblock_limb_can_merge(int C, int rows);
const char* resblock_limb_name(int C, int rows, bool merge = false);


// One ResBlock1 unit per branch for the wide first stage (C = 256, <= 32 rows per stream and step), a PAIR of workgroups per
// (branch, stream) tile: c1 split over output columns, c2 over input channels, partial sums exchanged at the end of the tile;
// the k-1 rows of xt in front of a tile come from a per-unit history ring (resblock_pair.hip).
struct RPProb {
  const float* w1; const float* w2;   // fragment-major like RBProb: [16 column tiles][k + 1 taps][16 K groups][64 lanes][4]
  const float* b1; const float* b2;   // [256]
  TRef x, y;                          // raw input / raw output
  TRef xh;                            // history of leaky_relu(c1 + b1): a ring with >= k - 1 rows of history, same rate as x
  int k, dil;
};
struct RPArgs {
  RPProb p[3];
  const int* slots; const int* pos;
  const int* tiles; int ntiles;       // filled by launch_resblock_pair
  int* sched;                         // queue state, 2 ints, zero before the first launch (re-armed by the kernel)
  float* xb;                          // exchange buffers, resblock_pair_xb_floats()
  unsigned* xflag; unsigned* mbox; unsigned* xcount;   // [pairs][8] flags, [pairs][4] tile mailboxes, [pairs][2] tile counts: zero at creation
  int nprob, n, T;                    // branches, slots, rows per slot (<= 32)
  float slope;
  unsigned* guard;                    // SpinGuard block of the stream-set (nullptr: unbounded waits)
  int fault;                          // test hook (conan_streams_test_fault): member 1 never posts its partial sums' flags
};
bool resblock_pair_supported(int C, int kmax, int span_max, int T);
size_t resblock_pair_xb_floats(int num_cu);
bool launch_resblock_pair(const RPArgs& a, int num_cu, hipStream_t st);
const char* resblock_pair_name(int T);

// Frame-rate conv / linear of the decoder step with an optional LayerNorm in front (rowconv.hip).
struct RowConvArgs {
  TRef x;               // input rows; with ln: the raw rows of THIS step (earlier rows come from `hist`)
  TRef hist;            // ln only: the layer's ring of normalised rows (read for t < 0, written for t >= 0)
  TRef y, res, m1, m2, lnmask, mask_out;
  const float* w;       // fragment-major: [Cout_pad/16][ktaps + 1][Cin/16][64 lanes][4]
  const float* bias;    // [Cout_pad] or nullptr
  const float* bvec; long long bvec_stride;   // per-slot broadcast vector or nullptr
  const float* gamma; const float* beta; float eps;
  const int* slots; const int* pos;
  int ln, has_res, has_m1, has_m2, has_lnmask, has_mask_out;
  int Cin, Cout, Cout_pad, ktaps, dil, T, n;
  int out_act; float out_scale, out_slope;
  int in_lrelu; float in_slope;   // LeakyReLU applied to the input window (HiFi-GAN resblock convs read raw tensors)
  int wr_max;           // filled by launch_rowconv: window rows of a tile
  int cw;               // filled by launch_rowconv: window channels (0: all of Cin; 1x1 layers wider than 512: chunks of 512)
  // ---- decoder megakernel only (decoder_mega.hip)
  // MOP_FFN: this conv (k = 1, Cout = hidden width, activation applied) is followed IN the workgroup by a second 1x1 conv
  // hidden -> Cout2 whose K range is the member's own hidden columns: member s of the group writes the partial sums
  // part[s][row][Cout2]; bias, residual and everything behind them are applied where the sum is consumed (xp.. below)
  const float* w2;      // second conv, fragment-major [Cout2_pad/16][2][Cout/16][64][4]
  float* part; long long part_stride;      // floats between the members' partial tensors, each [n*T][Cout2]
  int Cout2, Cout2_pad;
  // consumer side: the NEW rows of x are not a tensor but xparts partial tensors: x = ((p0 + p1 + ..) + xbias) + xres
  const float* xp; long long xp_stride; int xparts, xp_ld;
  const float* xbias; TRef xres; int has_xres;
  // ... and, for the decoder's fused conv blocks (LN -> k5 conv -> GELU -> 1x1 partial sums, conv.py:127-264), the rest of the
  // producer's epilogue: x = (((p0 + p1 + ..) + xbias) + xres) * xm1 * xm2; xstore: the summed rows are also stored to `x`
  // (row r by member r % members), where the NEXT fused operator's consumer finds them as its xres
  TRef xm1, xm2; int has_xm1, has_xm2, xstore;
  int hid_overlay;      // MOP_FFN with one 64-column hidden strip per member: the hidden tile overlays the window (read out by then)
};
bool rowconv_supported(int Cin, int ktaps, int dil, int T);
// plan_rows (all three): the row count the kernel variant is chosen for; 0 = the launch's own n * T (rowconv.hip)
void launch_rowconv(const RowConvArgs& a, hipStream_t st, int plan_rows = 0);
const char* rowconv_kernel_name(const RowConvArgs& a, int plan_rows = 0);   // as rocprofv3 prints it
int rowconv_plan(RowConvArgs& a, int* nbx, int* nby, int* lds_floats, int plan_rows = 0);   // tile geometry for the decoder megakernel

// LayerNorm over the channel axis of each row:
//   y[i][t][:] = (LN(x[i][t][:] (+ pre[i][t][:])) * gamma + beta) * m1 * m2 (+ post[i][t][:])
// optionally writing mask_out[i][t] = (sum_c |x| > 0).
struct LNArgs {
  TRef x, pre, y, post, m1, m2, mask_out;
  const float* gamma;
  const float* beta;
  const int* slots;
  const int* pos;
  const int* lens;
  int has_pre, has_post, has_m1, has_m2, has_mask_out;
  int T, n, C;
  float eps;
  // decoder megakernel only: x is the sum of xparts partial tensors (+ xbias, + xres) - see RowConvArgs
  const float* xp; long long xp_stride; int xparts, xp_ld;
  const float* xbias; TRef xres; int has_xres;
};
void launch_layernorm(const LNArgs& a, hipStream_t st);

// rows copy / gather helpers
struct CopyArgs {
  TRef x, y;
  const int* slots; const int* pos; const int* lens;
  int T, n, C;
};
void launch_copy_rows(const CopyArgs& a, hipStream_t st);

struct EmbedArgs {       // y[i][t][:] = table[idx[i][t]][:]
  TRef y;
  const float* table; const int* idx;
  const int* slots; const int* pos;
  int T, n, C, vocab;
};
void launch_embed(const EmbedArgs& a, hipStream_t st);

// Emformer attention for one layer (torchaudio _EmformerAttention.infer):
// queries = R+U tokens (+ the summary token when M > 0); keys = valid memory entries (min(M, ceil(past/seg))) | rc(R) |
// cached left context (min(LC, past)) | utt(U); the summary query does not see the memory columns; then the U new
// utterance keys/values are appended to the per-slot rings.
struct EmfAttnArgs {
  const float* q;      // [n][R+U(+1)][D]  (emb_to_query output, unscaled; last row = summary query when M > 0)
  const float* kv;     // [n][M+R+U][2D]   (emb_to_key_value output; rows [0,M) = memory bank entries, right-aligned)
  float* out;          // [n][R+U(+1)][D]
  float* kring; float* vring;   // [slot][LR][D]
  long long ring_slot_stride;
  const int* slots; const int* past;   // past[slot]
  int n, R, U, D, H, LC, lmask;
  float scaling;
  int M, seg;          // memory bank size (0: none) and segment length
};
void launch_emf_attn(const EmfAttnArgs& a, hipStream_t st);

// Memory-bank bookkeeping of one Emformer layer (torchaudio _EmformerLayer._unpack_state / _pack_state, memory_op):
//   ln[i][M+R+U]   = mean over the U utterance rows of ln (the summary token)
//   ln[i][0..M)    = the bank's last min(M, ceil(past/seg)) entries, right-aligned, zeros before them
//   bank[slot]    <- append mems_in[i]   (entry of segment number ceil(past/seg); ring of MB >= M+1 rows)
struct EmfMemArgs {
  float* ln;            // [n][M+R+U+1][D]
  float* bank;          // [slot][MB][D]
  const float* mems_in; // [n][D]
  const int* slots; const int* past;
  int n, R, U, D, M, MB, seg;
};
void launch_emf_mem_prep(const EmfMemArgs& a, hipStream_t st);
// out[i][:] = mean over rows [row0, row0+U) of x[i] ([n][rows][D])
void launch_emf_seg_mean(const float* x, float* out, int n, int rows, int row0, int U, int D, hipStream_t st);
// out[i][:] = tanh_on_mem ? tanh(x[i][row][:]) : clamp(x[i][row][:], -10, 10)
void launch_emf_mem_out(const float* x, float* out, int n, int rows, int row, int D, int tanh_on_mem, hipStream_t st);

// Whole streaming Emformer step in one launch (emformer_fused.hip): all layers + projection + arg-max.
constexpr int EMF_MAX_LAYERS = 12;
struct EmfLayerW {
  // fragment-major weights (ctx.hip finalize_emformer): 1 KiB per (16-column tile, 16-row k-group) MFMA B operand
  const float *wqkv, *wo, *w1, *w2;
  // [bq D | bkv 2D | bo D | b2 D | ln_in g,b | ln_ff g,b | ln_out g,b | b1 F]: one contiguous block per layer
  const float* params;
};
struct EmfFusedArgs {
  EmfLayerW layers[EMF_MAX_LAYERS];
  float* kring[EMF_MAX_LAYERS]; float* vring[EMF_MAX_LAYERS];          // [slot][LR][D]
  long long ring_slot_stride;
  const float* chunk;      // [n][U+R][D]  (utterance rows first, then the right context)
  float* out;              // optional [n][U][D]
  float* logits;           // optional [n][U][K]
  int* codes;              // optional [n][U]
  const float* wp; const float* bp;    // output projection, fragment-major (nullptr when output_dim == input_dim)
  const int* slots; int* past;         // past[slot] is advanced by U at the end of the launch
  int n, L, R, U, D, H, LC, lmask, F, K;
#ifdef EF_STAMPS
  unsigned long long* dbg;
#endif
  unsigned magic_per_g;   // ceil(2^32 / (max(LC,1) * D/4)): exact e / per_g for the prefetch units
  float scaling;
  // cluster mode (cs = 2, 4, 8): cs workgroups share one group of streams, each runs 1/cs of the feed-forward hidden
  // units and the partial sums meet in `xch` once per layer.  Workspace sizes: emformer_cluster_ws().
  int cs;
  float* xch;             // [cluster][2][EMF_MAX_CHUNKS][16][D] the feed-forward's per-chunk partial sums (layer parity double buffer)
  unsigned* xflag;        // [cluster][EMF_MAX_LAYERS][EMF_MAX_CLUSTER] "partial of this launch is written" (= epoch + 1)
  unsigned* xepoch;       // [cluster] launches this cluster has taken part in
  int fenced;             // 1: release / acquire fences around the exchange as well (CONAN_FENCED=1 cross-check)
  unsigned* guard;        // SpinGuard block of the stream-set (nullptr: unbounded waits)
  unsigned fault;         // test hook (conan_streams_test_fault): the cluster waits for a flag value nobody writes
  // memory bank (max_memory_size = M > 0): per layer the PROJECTED key / value rows of the last memory inputs, rings of MB >= M + 1
  // (power of two) entries per slot; entry of segment number j in row j % MB
  int M, MB, tanh_on_mem;
  float* bank_k[EMF_MAX_LAYERS]; float* bank_v[EMF_MAX_LAYERS];
  long long bank_slot_stride;
};
constexpr int EMF_MAX_CLUSTER = 8;
constexpr int EMF_MAX_CHUNKS = 16;      // hidden chunks of 256 columns per layer (ffn_dim <= 4096)
// floats of xch / words of xflag+xepoch for up to `max_groups` stream groups
inline size_t emformer_cluster_xch_floats(int max_groups, int D) { return (size_t)max_groups * 2 * EMF_MAX_CHUNKS * 16 * D; }
inline size_t emformer_cluster_flag_words(int max_groups) { return (size_t)max_groups * (EMF_MAX_LAYERS * EMF_MAX_CLUSTER + 1); }
int emformer_fused_streams_per_block(const EmfFusedArgs& a);
bool emformer_fused_supported(const EmfFusedArgs& a);
void launch_emformer_fused(const EmfFusedArgs& a, hipStream_t st);

// Cross attention of the prosody aligner (nn.MultiheadAttention, 2 heads) against cached K/V.
struct XAttnArgs {
  TRef q;              // [i][t][E]   already scaled by 1/sqrt(dh)
  TRef out;            // [i][t][E]
  const float* kv;     // [slot][S_max][2E]  (K | V)
  long long kv_slot_stride;
  const float* kmask;  // [slot][S_max] 0 or -inf
  const int* slen;     // [slot]
  float* attn_avg;     // optional [i][t][S_max] head-averaged weights
  const int* slots; const int* pos;
  int T, n, E, H, S_max;
};
void launch_xattn(const XAttnArgs& a, hipStream_t st);

// uv/f0 head: LN(128) -> Linear(128->2) -> uv/f0 -> coarse bin -> decoder_inp = pitch_inp + pitch_embed[bin]
struct PitchHeadArgs {
  TRef h;              // [i][t][Cp]
  TRef pitch_inp;      // [i][t][E]
  TRef dec_inp;        // [i][t][E]
  const float* gamma; const float* beta; const float* w; const float* b;   // w [2][Cp]
  const float* pitch_embed;     // [300][E]
  const int* codes;             // [n][T]
  float* uv_pred; float* f0; int* bins;   // optional taps, [n][T][2], [n][T], [n][T]
  const int* slots; const int* pos;
  int T, n, Cp, E, silent_token;
};
void launch_pitch_head(const PitchHeadArgs& a, hipStream_t st);

// y = leaky_relu((x0 + x1 + x2) / nsrc): the MRF mean of HifiGanGenerator.forward (hifigan_causal.py:324-331) with the
// following LeakyReLU, materialised once so that the consuming conv runs the single-source direct-to-LDS path.
struct MeanActArgs { TRef x[3]; TRef y; const int* slots; const int* pos; int nsrc, T, n, C; float slope; };
void launch_mean_act(const MeanActArgs& a, hipStream_t st);
// conv_post (CausalConv1d(C -> 1, k) + tanh, hifigan_causal.py:331-333) as a VALU dot-product kernel: N = 1 would
// waste 31/32 of an MFMA tile.  x[0] is the already activated input ring (nsrc = 1); w is [k][C]; optional pre-tanh tap.
// xmean.base != nullptr: x[0 .. nsrc) are the RAW outputs of the last stage's branches (nsrc may be 1: a single-branch
// vocoder still needs its LeakyReLU in front of conv_post) and the kernel forms leaky_relu(mean) itself (the same
// operations in the same order as mean_act_kernel), so the stage's own mean_act launch disappears; rows of earlier steps
// (the k - 1 rows of left context) come from `xmean`, the activated-mean ring, to which the kernel also appends the rows
// it formed - the ring stays valid whichever way a step produced it (merged fused launch, or here).
// adv_pos != nullptr: the last workgroup to finish advances the per-slot frame counters by adv_delta (the step's
// launch_advance folded in; adv_ticket is a zeroed int the kernel leaves zeroed).
struct ConvPostArgs { TRef x[3]; TRef xmean; int nsrc; float slope; const float* w; float bias; float* wav; float* pre; const int* slots; const int* pos; int T, n, C, k;
                      int* adv_pos; int adv_delta; int* adv_ticket; };
void launch_conv_post(const ConvPostArgs& a, hipStream_t st);

// The decoder step as ONE persistent launch (decoder_mega.hip): a list of row-wise operators (rowops.h) walked by every
// workgroup, tiles of an operator dealt round-robin over the grid, a grid barrier between dependent operators.
enum MegaOpType { MOP_RC111 = 0, MOP_RC114 = 1, MOP_ROWLIN = 2, MOP_LN = 3, MOP_XATTN = 4, MOP_PITCH = 5, MOP_EMBED = 6, MOP_COPY32 = 7, MOP_ADVANCE = 8, MOP_FFN = 9 };
struct MegaCopy { unsigned* dst; const unsigned* src; long long n; };          // n 32-bit words (plain accesses: inputs of the launch -> outputs read after it)
struct MegaAdvance { int* pos; const int* slots; int n, delta; };              // pos[slots[q]] += delta (the step's last operator)
struct MegaOp {
  int type, nbx, nby;
  int barrier;            // 1: the operators after this one read what it wrote - grid barrier behind it
  float f0, f1;           // pitch head: mel_min, mel_max - mel_min
  int pad_[2];
  union U { RowConvArgs rc; LNArgs ln; XAttnArgs xa; PitchHeadArgs ph; EmbedArgs em; MegaCopy cp; MegaAdvance adv; } u;
};
int decoder_mega_lds_floats(const MegaOp& op, int rc_lds_floats);
int decoder_mega_blocks_per_cu(int lds_bytes, bool wide_regs);
// prog: device copy of nops operators.  njobs 16-row tiles are dealt over `groups` groups of `group_size` workgroups; kw4: the
// step is one tile and its strips are 16 columns with the K loop split over a workgroup's waves.  gbar: one zero-initialised
// counter per group, 16 words apart; bar: the grid barrier's counter, counts for ever - bar_base is its value before this
// launch (the host adds groups * group_size per launch).
struct MegaLaunch {
  const MegaOp* prog; int nops, njobs, groups, group_size, kw4, lds_bytes;
  const int* slots; const int* pos; int n, T;
  unsigned* gbar; unsigned* bar; unsigned bar_base; unsigned long long* dbg;
  unsigned* guard;        // SpinGuard block of the stream-set (nullptr: unbounded waits)
  int wide_regs;          // 1: the 128-register build (bf16-limb stream-sets), 0: the 80-register build
  // xcd mode (a single row tile in the step): one workgroup per CU is launched, those on workgroup 0's XCD form the one group
  int xcd; unsigned* xs; unsigned xseq, xdec_base;
};
void launch_decoder_mega(const MegaLaunch& m, hipStream_t st);

struct ArgmaxArgs { const float* x; int* idx; int rows, C; };
void launch_argmax(const ArgmaxArgs& a, hipStream_t st);

void launch_advance(int* pos, const int* slots, int n, int delta, hipStream_t st);
void launch_fill_int(int* p, const int* slots, int n, int value, hipStream_t st);
void launch_copy_int_rows(int* dst, const int* src, int n, int T, int S, hipStream_t st);
void launch_profile_mark(hipStream_t st);
void launch_scatter_int(int* dst, const int* slots, const int* src, int n, hipStream_t st);
void launch_scatter_ids(int* dst, const int* slots, const int* src, const int* lens, int n, int S_max, hipStream_t st);   // dst[slot][s] = s < lens[i] ? src[i][s] : -1
void launch_gather_ids(int* dst, int* cnt, const int* src, const int* slen, const int* slots, int n, int S_max, hipStream_t st);
void launch_scatter_rows(float* dst, const float* src, const int* slots, int n, int C, hipStream_t st);   // dst[slots[i]][:] = src[i][:]
void launch_zero_slots(float* base, long long slot_stride, long long count, const int* slots, int n, hipStream_t st);

// ---- style pass (per utterance) helpers
struct RowMaskArgs { TRef x; TRef m; const int* lens; int T, n, C; int mode; };  // mode 0: sum|x|>0, 1: x[0]!=0
void launch_rowmask(const RowMaskArgs& a, hipStream_t st);
struct WNGateArgs { TRef xin; TRef acts; const int* lens; int T, n, H; };          // tanh(a[:H])*sigmoid(a[H:])
void launch_wn_gate(const WNGateArgs& a, hipStream_t st);
struct WNUpdateArgs { TRef rs; TRef x; TRef out; TRef m; const int* lens; int T, n, H; int last; int first; };
void launch_wn_update(const WNUpdateArgs& a, hipStream_t st);
struct PoolArgs { TRef x; TRef m; TRef y; const int* lens; int T, n, C, group; };  // y = mean over groups of x*m
void launch_group_pool(const PoolArgs& a, hipStream_t st);
struct VQArgs {       // dots[i][s][M] = x.e ; picks argmin of (e2 + x2) - 2*dot, writes z = x + (q - x) and [z | posemb]
  TRef x; TRef dots; TRef cat;
  const float* emb; const float* e2; const float* postable;
  int* ids;            // [i][S_max]
  const int* lens; int S, n, E, M, S_max;
};
void launch_vq(const VQArgs& a, hipStream_t st);
struct KMaskArgs { TRef tok; float* kmask; long long kmask_stride; const int* slots; const int* lens; int S, n, S_max; };
void launch_kmask(const KMaskArgs& a, hipStream_t st);
struct MeanArgs { TRef x; TRef m; float* out; long long out_stride; const int* slots; const int* lens; int T, n, C; };
void launch_masked_mean(const MeanArgs& a, hipStream_t st);
struct ScaleMaskArgs { TRef x; TRef m; const int* lens; int T, n, C; };           // x *= m (in place)
void launch_mul_mask(const ScaleMaskArgs& a, hipStream_t st);

}  // namespace cnk
