// conan_streams: per-slot streaming state and launch plans (see streams.hip, decoder.hip, style.hip).
#pragma once
#include "host_common.h"

using ch::Error;
using ch::Lin;
using ch::PackedConv;
using ch::Ring;
using cnk::ConvArgs;
using cnk::ConvGroup;
using cnk::TRef;

constexpr int kMaxBranches = 3;
constexpr int PADR = 16;   // zero rows before/after a reference utterance in the style-pass buffers (k31 -> 15)

struct VocStage {
  Ring up;                                  // ups[i] output (pixel shuffled), also residual source
  Ring upa;                                 // leaky_relu(up): c1 operand
  Ring xs;                                  // leaky_relu(mean of the branches): input of ups[i+1] / conv_post
  std::vector<std::vector<Ring>> xt, xo;    // [branch][dilation]; xt is stored activated
  std::vector<std::vector<Ring>> xa;        // leaky_relu(xo) for the outputs that feed another c1
  int C = 0, rate = 1;
  bool fused = false;                       // ResBlock1 units run as one tile pass each (resblock_fused.hip): no xt / activated twins
  bool pair = false;                        // ... by pairs of workgroups (resblock_pair.hip: the wide first stage); xh = history of activated xt per unit
  std::vector<std::vector<Ring>> xh;
};

// Small host tables (slot lists, reference lengths) go to the device through a ring of pinned staging buffers: an
// asynchronous copy from pageable memory may still be reading the host buffer after the call returns, and the callers'
// vectors do not live that long.  A buffer is reused only after the copy that read it has completed (event).
struct PinRing {
  static constexpr int N = 8;
  int* buf = nullptr;
  size_t cap = 0;            // ints per buffer
  hipEvent_t ev[N] = {};
  int next = 0, cur = 0;
  void init(size_t ints) {
    cap = ints;
    HIP_CHECK(hipHostMalloc((void**)&buf, cap * N * sizeof(int), hipHostMallocDefault));
    for (int i = 0; i < N; ++i) HIP_CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
  }
  // copy `n` ints to `dst` (device) on `st`
  void upload(int* dst, const int* src, size_t n, hipStream_t st) {
    if (n > cap) throw ch::Error(CONAN_ERR_INVALID, "host table larger than the staging buffer");
    cur = next; next = (next + 1) % N;
    HIP_CHECK(hipEventSynchronize(ev[cur]));          // never recorded / long complete in the steady state
    memcpy(buf + (size_t)cur * cap, src, n * sizeof(int));
    HIP_CHECK(hipMemcpyAsync(dst, buf + (size_t)cur * cap, n * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_CHECK(hipEventRecord(ev[cur], st));
  }
  ~PinRing() {
    for (int i = 0; i < N; ++i) if (ev[i]) (void)hipEventDestroy(ev[i]);
    if (buf) (void)hipHostFree(buf);
  }
};

struct conan_streams {
  conan_ctx* ctx = nullptr;
  std::atomic<int>* live = nullptr;            // this device's count of live stream-sets (device_live_streams)
  // developer / test switches of the launch plan: conan_streams_opts.dev_plan ("NAME=value;..."; DEV builds: also CONAN_<NAME> in the environment)
  std::map<std::string, std::string> dev_plan;
  void parse_dev_plan(const char* text);
  const char* dev(const char* name) const {
    if (!dev_plan.empty()) {
      auto it = dev_plan.find(name);
      if (it != dev_plan.end()) return it->second.c_str();
    }
#ifdef CONAN_DEV_SWITCHES
    return getenv((std::string("CONAN_") + name).c_str());
#else
    return nullptr;
#endif
  }
  // CONAN_STREAMS_FIXED_PLAN: every plan choice from max_slots, never from the step's active slot count
  bool fixed_plan = false;
  // (the per-utterance style pass of a fixed-plan stream-set runs one slot at a time: its plan follows the slot's own reference length)
  bool in_style_pass = false;
  int plan_n(int n) const { return (fixed_plan && !in_style_pass) ? max_slots : n; }
  bool pipe_idle = false;                      // set by conan_step_async: no earlier pipelined step of this stream-set is still in flight
  bool shared_device = false;                  // CONAN_STREAMS_SHARED_DEVICE: other processes drive this GPU too - never take whole-chip launch shapes
  int max_slots = 0, max_frames = 0, max_ref = 0, S_max = 0;
  std::vector<void*> allocs;
  int64_t state_bytes = 0;
  std::vector<std::pair<float*, long long>> voc_state, dec_state, emf_state;  // (base, floats per slot) to zero on reset

  int* d_slots = nullptr;   // [max_slots]
  int* d_ident = nullptr;   // identity 0..max_slots-1
  int* d_zero = nullptr;    // zeros (pos array for batch-indexed style pass)
  int* d_lens = nullptr;    // [max_slots] per-batch lengths (style pass)
  int* d_lens2 = nullptr;
  int* d_codes = nullptr;   // [max_slots][max_frames] codes scratch for the fused step
  // inter-block split-K workspaces (partial tiles + ticket counters), one per stream that can have conv launches in
  // flight (ws_index): the caller's stream / pipelined decoder, the pipelined vocoder, the pipelined Emformer
  // (conan_step_async runs the three concurrently, and all split K at small batch sizes)
  float* sk_slab[3] = {nullptr, nullptr, nullptr};
  int* sk_counters[3] = {nullptr, nullptr, nullptr};
  int reserve_cus = 0;                     // CUs the pipelined vocoder's persistent launches leave to the front-end stream (CONAN_RESERVE_CUS)
  bool fenced = false;          // CONAN_FENCED=1 at creation: release / acquire fences around the inter-workgroup hand-offs too
  bool rb_limb = false;         // bf16-limb form of the vocoder's matrix kernels where it exists (conan_streams_opts.arith, resolved at creation)
  bool arith_auto = true;       // the caller left the choice to the library (conan_streams_opts.arith == AUTO)
  bool rb_merge = true;         // merged-branch last-dilation launches (CONAN_RB_NOMERGE=1 at creation: separate branches + mean_act)
  int* cp_ticket[3] = {nullptr, nullptr, nullptr};   // conv_post's last-workgroup ticket, per internal stream
  int* rb_sched[2] = {nullptr, nullptr};   // work-queue counters of the fused resblock launches, per stream like the split-K workspaces
  // resblock_pair workspaces (per stream like rb_sched): exchange buffers, and one block of zero-initialised words:
  // [pairs][8] flags | [pairs][4] mailboxes | [pairs][2] tile counts | 2 queue words
  float* rp_xb[2] = {nullptr, nullptr};
  unsigned* rp_words[2] = {nullptr, nullptr};
  long long sk_slab_floats = 0;
  int sk_max_tiles = 0;
  std::vector<int> h_slots;
  std::vector<int> slot_seen;   // duplicate detection: generation stamp per slot
  int slot_gen = 0;
  PinRing pin;
  int* pos_emf = nullptr; int* pos_dec = nullptr; int* pos_voc = nullptr;
  // bounded cross-workgroup waits (kernels.h, SpinGuard): guard block in device memory + the host-mapped word a waiter that gave
  // up copies its code to; check_fault() turns a non-zero word into CONAN_ERR_HIP at every stream-ordered entry point, for good
  unsigned* d_guard = nullptr;
  unsigned* h_guard = nullptr;
  int test_fault = 0;           // conan_streams_test_fault: the next launch of that kind waits for an arrival that never comes
  void check_fault() const {
    if (!h_guard) return;
    const unsigned code = *(volatile const unsigned*)h_guard;
    if (code == 0) return;
    static const char* what[] = {"?", "decoder_mega_kernel group / grid barrier", "emformer_fused_kernel cluster exchange", "resblock_pair_kernel partner flag", "resblock_pair_kernel tile mailbox",
                                 "(retired)", "decoder_mega_kernel xcd election", "decoder_mega_kernel xcd roll call", "decoder_mega_kernel xcd flag barrier"};
    throw ch::Error(CONAN_ERR_HIP, std::string("a cross-workgroup wait gave up after its 50 ms budget (") + what[code < 9 ? code : 0] +
                                       "): results since then are invalid and this stream-set is unusable - destroy it and create a new one");
  }

  // --- vocoder
  Ring v_mel, v_pre;
  std::vector<VocStage> v_st;
  // --- emformer
  std::vector<Ring> e_k, e_v;
  std::vector<float*> e_bank;     // memory bank per layer: [slot][e_bank_rows][D] (max_memory_size > 0)
  int e_bank_rows = 0;
  float* e_mems[2] = {nullptr, nullptr};   // memory input / output of a layer, [n][D]
  Lin e_x[2], e_ln, e_q, e_kv, e_att, e_r1, e_ffn, e_h, e_r2, e_logits;
  bool emf_fused = false;
  int emf_cluster = 0;          // workgroups per stream group of the fused step (0: per launch; CONAN_EMF_CLUSTER)
  cnk::EmfFusedArgs emf_fused_args;
  // --- conan decoder
  Ring c_emb, c_pin2, c_lastr;
  std::vector<Ring> c_uvh;      // outputs of the uv predictor's conv layers but the last (each keeps its own left context)
  std::vector<Ring> c_lnrs;     // post-LN rings, one per (block, sub-layer)
  Lin c_pin, c_q, c_att, c_a1, c_a2, c_ff, c_uv5, c_x[2], c_h, c_post, c_mask_blk, c_mask_blk2, c_mask_out, c_mel, c_part, c_part2;
  float* c_style = nullptr;     // [slot][H]
  float* c_kv = nullptr;        // [slot][2 layers][S_max][2H]
  float* c_kmask = nullptr;     // [slot][S_max]
  int* c_slen = nullptr;        // [slot]
  int* c_vqids = nullptr;       // [slot][S_max] VQ indices of the prosody tokens (-1 past the token count)
  std::vector<char> has_ref;    // per slot: conan_set_reference has run for it
  std::vector<char> voc_fresh;  // per slot: vocoder state reset and not stepped since (voc_upsample 2 steps need it)
  // --- style pass workspace (batch indexed, max_slots_sp at a time)
  int sp_batch = 0;
  Lin s_mel, s_np, s_wnm, s_x[2], s_ln, s_h, s_blkm, s_wx, s_wout, s_win, s_acts, s_rs, s_ph, s_pm, s_px[2], s_pln, s_phh,
      s_pblk, s_enc, s_dots, s_cat, s_tok, s_kvtmp;
  int* s_ids = nullptr;

  float* alloc(size_t floats) {
    void* p = nullptr;
    if (floats == 0) floats = 4;
    HIP_CHECK(hipMalloc(&p, floats * sizeof(float)));
    HIP_CHECK(hipMemset(p, 0, floats * sizeof(float)));
    allocs.push_back(p);
    state_bytes += (int64_t)floats * 4;
    return (float*)p;
  }
  Ring mk_ring(int C, int rate, int hist, std::vector<std::pair<float*, long long>>* reg) {
    Ring r; r.C = C; r.rate = rate;
    r.L = ch::next_pow2(hist + max_frames * rate);
    r.slot_stride = (long long)r.L * C;
    r.base = alloc((size_t)max_slots * r.slot_stride);
    if (reg) reg->push_back({r.base, r.slot_stride});
    return r;
  }
  Lin mk_lin(int rows, int C, int nb = -1) {
    Lin l; l.rows = rows; l.C = C;
    l.base = alloc((size_t)(nb < 0 ? max_slots : nb) * rows * C);
    return l;
  }
  // --- pipelined stepping (conan_step_async): front-end (Emformer + decoder) and vocoder on two internal streams
  hipStream_t st_emf = nullptr, st_front = nullptr, st_voc = nullptr;
  // pipelined steps: recorded on the vocoder stream behind the wide first stage's pair-kernel launches (its workgroups wait for
  // their partners: a CU that an Emformer workgroup holds stalls a whole pair).  Developer switch CONAN_EMF_HOLD=1: the Emformer of
  // step t is held back until the vocoder of step t-2 has passed that point.  Measured: with the limb kernels and the pair kernel
  // (CONAN_RB_PAIR=1) the mean step is unchanged and the p95 of the step intervals falls from 1.63 to 1.57 ms; with the exact-f32
  // kernels (1.77 ms steps) it costs 2.6 % (1.815 against 1.769 ms) - off by default.
  hipEvent_t ev_wide[4] = {};
  hipEvent_t mark_wide = nullptr;            // set around hifigan_step by conan_step_async
  bool wide_marked[4] = {false, false, false, false};   // the step at this ring position recorded its ev_wide (it has a pair stage)
  static constexpr int NP = 4;                 // depth of the hand-off rings: a stage may run up to NP steps ahead of its consumer
  hipEvent_t ev_in[NP] = {}, ev_emf[NP] = {}, ev_front[NP] = {}, ev_voc[NP] = {};   // (one input event per ring position: a single
                                               // event re-recorded while its previous record is still pending stalls the pipeline)
  hipEvent_t ev_fence[NP] = {};                // output fences (conan_streams_output_fence)
  hipStream_t fence_stream = nullptr; bool fence_set = false;
  hipEvent_t fence_event = nullptr;            // conan_streams_output_fence_event: wait for this recorded event instead of the stream's tail
  int* codes_hand[NP] = {};                    // code hand-off buffers Emformer -> decoder [max_slots][segment]
  // workspace index of a stream: 0 caller / pipelined decoder, 1 pipelined vocoder, 2 pipelined Emformer
  int ws_index(hipStream_t st) const { return (st_voc && st == st_voc) ? 1 : ((st_emf && st == st_emf) ? 2 : 0); }
  float* mel_hand[NP] = {};                    // mel hand-off buffers decoder -> vocoder [max_slots][max_frames][num_mels]
  long long async_steps = 0;                   // steps enqueued since creation
  std::vector<hipEvent_t> clock_ev;            // conan_step_clock: one timing event per pipelined step, on the vocoder stream
  bool clock_on = false; int clock_n = 0;
  std::vector<hipEvent_t> tl_ev;               // conan_step_timeline: 6 timing events per pipelined step (start / end of each stage on its stream)
  bool tl_on = false; int tl_n = 0;
  void async_init();
  void join(hipStream_t st);                   // make `st` wait for everything enqueued by conan_step_async

  void mega_print_stamps();
  ~conan_streams() {
    if (live) live->fetch_sub(1);
    if (mega_dbg) mega_print_stamps();
    if (st_emf) (void)hipStreamDestroy(st_emf);
    if (st_front) (void)hipStreamDestroy(st_front);
    if (st_voc) (void)hipStreamDestroy(st_voc);
    for (int i = 0; i < NP; ++i) { if (ev_in[i]) (void)hipEventDestroy(ev_in[i]); if (ev_fence[i]) (void)hipEventDestroy(ev_fence[i]); }
    for (int i = 0; i < NP; ++i) { if (ev_wide[i]) (void)hipEventDestroy(ev_wide[i]); if (ev_emf[i]) (void)hipEventDestroy(ev_emf[i]); if (ev_front[i]) (void)hipEventDestroy(ev_front[i]); if (ev_voc[i]) (void)hipEventDestroy(ev_voc[i]); }
    for (void* p : allocs) (void)hipFree(p);
    if (h_guard) (void)hipHostFree(h_guard);
    for (auto& e : prof_ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto& e : clock_ev) (void)hipEventDestroy(e);
    for (auto& m : mega_cache) { if (m.copied) (void)hipEventDestroy(m.copied); if (m.pinned) (void)hipHostFree(m.pinned); if (m.dev) (void)hipFree(m.dev); }
    for (auto& e : tl_ev) (void)hipEventDestroy(e);
  }

  void build_vocoder();
  void build_emformer();
  void build_decoder();
  void set_slots(const int32_t* slots, int n, hipStream_t st);
  int pick_cfg(int M, int N, int nprob) const;
  // optional per-launch timing of the conv kernel family (conan_profile_*): HIP events on the launch stream
  bool prof_on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev;
  size_t prof_used = 0;
  double prof_flops = 0.0;
  long long prof_launches = 0;
  struct ProfRec { std::string name; double flops; };
  std::vector<ProfRec> prof_rec;                 // one per recorded launch (same order as prof_ev)
  struct ProfKernel { std::string name; double ms, flops; long long n; };
  std::vector<ProfKernel> prof_kernels;          // filled by conan_profile_end: per template instantiation
  void launch_group(const ConvGroup& g, int nprob, int cfg, hipStream_t st);
  bool launch_rb(const cnk::RBArgs& a, int C, hipStream_t st, const TRef* ymean = nullptr);   // true: the launch stored the branch mean (merged)
  void launch_rp(const cnk::RPArgs& a, hipStream_t st);
  bool use_rowconv = true;                  // frame-rate decoder layers through rowconv.hip (CONAN_ROWCONV=0: conv_mfma + LayerNorm launches)
  bool rowconv_ok(const PackedConv& pc, int dil, int T) const { return use_rowconv && pc.wf && cnk::rowconv_supported(pc.Cin, pc.k, dil, T); }
  cnk::RowConvArgs mk_rc(const PackedConv& pc, const TRef& x, const TRef& y, int n, int T, int dil = 1) const;
  void rowconv(const cnk::RowConvArgs& a, hipStream_t st);
  template <typename F> void profiled(const std::string& name, double flops, hipStream_t st, F&& launch);
  void conv(const ConvArgs& a, hipStream_t st) { ConvGroup g; g.p[0] = a; launch_group(g, 1, pick_cfg(plan_n(a.n) * a.T, a.Cout, 1), st); }
  ConvArgs mk(const PackedConv& pc, const TRef& x, const TRef& y, int n, int T, const int* pos, int dil = 1, int pad_left = -1) const;

  // --- decoder megakernel (decoder_mega.hip): the decoder step's operator list, recorded once per (slot count, frames,
  // buffer set) and replayed as one persistent launch
  struct DecExtra { float* mel_out2 = nullptr; int* codes_dst = nullptr; const int* codes_src = nullptr; int codes_words = 0; };
  struct MegaProgram {
    long long key[6] = {0, 0, 0, 0, 0, 0};
    bool ok = false;
    int nops = 0, groups = 0, group_size = 0, njobs = 0, kw4 = 0, lds_bytes = 0, barriers = 0, n = 0, T = 0;
    int lds_need = 0;                  // what the operators need (lds_bytes may be padded: xcd mode, blocking steps)
    bool xcd = false;                  // a single row tile: the launch's workgroups on ONE XCD form the group (decoder_mega.hip)
    double flops = 0.0;
    cnk::MegaOp* dev = nullptr;        // device copy (capacity kMegaMaxOps)
    cnk::MegaOp* pinned = nullptr;     // host staging of this entry; reused only after `copied` has fired
    hipEvent_t copied = nullptr;
    long long stamp = 0;               // least-recently-used replacement
  };
  static constexpr int kMegaMaxOps = cnk::kMegaMaxOps, kMegaEntries = 12;
  std::vector<MegaProgram> mega_cache;
  long long mega_clock = 0;
  bool use_mega = true;                          // CONAN_DEC_MEGA=0: the decoder step as separate launches
  int mega_grid = 128;                           // CONAN_MEGA_GRID
  int mega_gs = 8;                               // workgroups per group (CONAN_MEGA_GS: 4, 8 or 16)
  bool mega_narrow_ksplit = true;                // narrow layers of multi-tile launches as K-split 16-column strips (CONAN_MEGA_NARROW=0: off)
  int mega_ffn_gs = 8;                           // members of a fused feed-forward while a program is recorded (run_mega)
  unsigned* mega_bar = nullptr;                  // the grid barrier's arrival counter (counts for ever); the group counters follow it, 16 words apart
  unsigned mega_bar_count = 0;                   // its value once every launch enqueued so far has finished
  unsigned* mega_x = nullptr;                    // xcd mode: election word, rank counter, "decided" counter, barrier flags (decoder_mega.hip)
  unsigned mega_gseq = 0;                        // multi-tile launches so far (epochs of their groups' flag barriers)
  unsigned mega_xseq = 0, mega_xdec = 0;         // launches in xcd mode so far (24 bits), the decided counter's value once they have all finished
  int opt_flags = 0;                             // conan_streams_opts.flags (+ the developer environment overrides)
  bool mega_single = true;                       // single-tile steps take the persistent launch (xcd mode); CONAN_MEGA_SINGLE=0: separate launches
  unsigned long long* mega_dbg = nullptr;        // CONAN_MEGA_STAMPS=1: per-operator clock stamps of the last launch (printed at destruction)
  int mega_dbg_prog = -1;                        // index into mega_cache (the vector may reallocate)
  std::vector<cnk::MegaOp>* mega_rec = nullptr;  // != nullptr: decoder_ops() records its operators instead of launching them
  bool mega_rec_ok = true; int mega_rec_lds = 0; double mega_rec_flops = 0.0;
  void mega_push(cnk::MegaOp& op, int lds_floats);
  bool run_mega(int n, int T, const int32_t* codes, float* mel_out, const DecExtra& ex, hipStream_t st);
  void launch_mega(MegaProgram& e, hipStream_t st);
  void decoder_ops(int n, int frames, const int32_t* codes, float* mel_out, const conan_decoder_taps& taps, hipStream_t st);
  void op_embed(const cnk::EmbedArgs& a, hipStream_t st);
  void op_ln(const cnk::LNArgs& a, hipStream_t st);
  void op_xattn(const cnk::XAttnArgs& a, hipStream_t st);
  void op_pitch(const cnk::PitchHeadArgs& a, hipStream_t st);
  void op_advance(int* pos, int n, int delta, hipStream_t st);

  void hifigan_step(int n, int frames, const float* mel_dev, float* wav_out, float* pre_tanh, hipStream_t st, const conan_hifigan_taps* taps = nullptr);
  void emformer_step(int n, const float* chunk, float* out, float* logits, int32_t* codes, hipStream_t st);
  void decoder_step(int n, int frames, const int32_t* codes, float* mel_out, const conan_decoder_taps& taps, hipStream_t st, const DecExtra* extra = nullptr);
  void set_reference(const int32_t* slots, int n, const float* ref, const int32_t* ref_len, int max_len, hipStream_t st);
  void conv_blocks_noncausal(const std::string& name, int nblocks, int k, int C, Lin* x, Lin& ln, Lin& h, Lin& blkm, const TRef& npm,
                             const int* lens, int n, int T, int& cur, hipStream_t st);
};

// every launch of the matrix kernels goes through here: between conan_profile_begin / _end it is bracketed by HIP
// events on its launch stream and booked under the kernel's name with its algorithmic FLOPs
template <typename F>
inline void conan_streams::profiled(const std::string& name, double flops, hipStream_t st, F&& launch) {
  if (!prof_on) { launch(); return; }
  if (prof_used == prof_ev.size()) {
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
    prof_ev.push_back({a, b});
  }
  auto& ev = prof_ev[prof_used++];
  HIP_CHECK(hipEventRecord(ev.first, st));
  launch();
  HIP_CHECK(hipEventRecord(ev.second, st));
  prof_flops += flops;
  prof_launches += 1;
  prof_rec.push_back({name, flops});
}


