// Host side of voc_chain.hip: which stream-sets take the one-launch vocoder step, the phase programs, the launch.
#include "streams.h"

namespace {
constexpr int kChainRows = 16;            // slots x frames of a chain stream-set
constexpr int kChainLdsMax = 120 * 1024;  // dynamic LDS of a workgroup (a decoder megakernel workgroup still fits beside it)
constexpr int kChainHdrFloats = 192;      // voc_chain.hip VC_HDR
constexpr int kChainPostRows = 256;       // voc_chain.hip VC_POST_ROWS
constexpr int kChainGridMax = 192;        // workgroups: the Emformer's (<= 64, whole CUs) fit beside them on 256 CUs
}  // namespace

const float* conan_ctx::chain_weight(const std::string& name) {
  std::lock_guard<std::mutex> lock(chain_mu);
  auto it = chain_w.find(name);
  if (it != chain_w.end()) return it->second;
  const ch::PackedConv& pc = conv(name);
  if (pc.Cin % 16) throw ch::Error(CONAN_ERR_UNSUPPORTED, "voc_chain: input channels must be a multiple of 16: " + name);
  float* d = dev_alloc(cnk::voc_chain_weight_floats(pc.Cout, pc.Cin, pc.k));
  cnk::launch_voc_chain_repack(d, pc.w, pc.Cout, pc.Cout_pad, pc.Cin, pc.Cin_alloc, pc.k, nullptr);
  HIP_CHECK(hipStreamSynchronize(nullptr));
  chain_w[name] = d;
  return d;
}

// A stream-set CAN take the chain when it is small (one mel row tile), the vocoder is the shuffle-upsampler / ResBlock1 generator the
// kernel covers, and the caller did not ask for the bf16-limb arithmetic (the chain computes on the f32 MFMA).
// MEASURED (round 5, MI355X, DESIGN.md §4 "voc_chain"): parity-green (9e-7 against the launch plans, ragged steps included) but NOT
// faster - 0.40-0.42 ms per one-stream step against 0.35 ms of the 28 launches, 0.70 against 0.44 ms at four streams: 30 dependent
// phases of ~13 us (hand-off through memory 3 + gather 2 + the job's MFMAs, 2.8 us at one CU for a k = 11 tile, + slice sums and
// stores 1.6 + skew) cost what the launches do.  So the launch plans stay the default; conan_streams_opts.flags
// CONAN_STREAMS_VOCODER_CHAIN (developer override CONAN_VOC_CHAIN=1) selects the chain.
bool conan_streams::chain_eligible(bool limb_requested) const {
  const conan_cfg& c = ctx->cfg;
  if (!(c.models & CONAN_MODEL_HIFIGAN) || limb_requested) return false;
  if (!(opt_flags & CONAN_STREAMS_VOCODER_CHAIN)) return false;
  if ((long long)max_slots * max_frames > kChainRows || max_slots > 16) return false;
  if (c.voc_upsample != 0 || c.voc_resblock == 2 || c.voc_num_resblocks > kMaxBranches) return false;
  if (2 + c.voc_num_ups * (1 + 2 * c.voc_rb_num_dil) > cnk::VC_MAX_PHASES || 1 + 2 * c.voc_num_ups > cnk::VC_MAX_TAPS) return false;
  if (c.num_mels % 16 || c.voc_initial_channel % 16) return false;
  int ch_ = c.voc_initial_channel;
  // every job's window must fit into LDS: one 16-row tile plus the left context of the longest conv of each width
  auto fits = [&](int Cin, int k, int dil, int rows) { return (size_t)(kChainHdrFloats + (rows + (k - 1) * dil) * (Cin + 8)) * 4 <= (size_t)kChainLdsMax; };
  if (!fits(c.num_mels, 7, 1, 16)) return false;
  for (int i = 0; i < c.voc_num_ups; ++i) {
    if (!fits(ch_, c.voc_up_kernels[i], 1, 16)) return false;
    ch_ /= 2;
    if (ch_ % 16) return false;
    for (int b = 0; b < c.voc_num_resblocks; ++b)
      for (int d = 0; d < c.voc_rb_num_dil; ++d) if (!fits(ch_, c.voc_rb_kernels[b], c.voc_rb_dilations[b][d], 16)) return false;
  }
  return true;
}

namespace {
struct Geo { int NRT, NCT, KS, spt, tps, tiles, ncg, njobs, lds_floats; };

// tile shape of a conv phase: problems share n, T, Cin, ncts; kmax / dmax = the longest branch
bool plan_phase(int n, int T, int Cin, int ncts, int nprob, int kmax, int halo_max, int G, Geo* out) {
  const int KQ = Cin / 16, LDX = Cin + 8;
  double best = 1e30; bool found = false;
  for (int NRT : {1, 2, 4}) {
    if (T < 16 && NRT > 1) continue;
    if (NRT > 1 && 16 * NRT > T) continue;
    int spt = 0, tps = 0, tiles = 0, WR = 0;
    if (T < 16) {
      spt = std::max(1, std::min(16 / T, n));
      while (spt > 1 && (size_t)spt * (halo_max + T) * LDX * 4 > (size_t)kChainLdsMax - 4096) --spt;
      tiles = (n + spt - 1) / spt; WR = spt * (halo_max + T);
    } else { tps = (T + 16 * NRT - 1) / (16 * NRT); tiles = n * tps; WR = halo_max + 16 * NRT; }
    for (int NCT : {1, 2, 4, 8}) {
      if (NCT > ncts || ncts % NCT) continue;
      const int KS = 8 / NCT;
      const int lds = kChainHdrFloats + WR * LDX + (KS > 1 ? 2048 * NRT : 0);
      if ((size_t)lds * 4 > (size_t)kChainLdsMax) continue;
      const long long njobs = (long long)nprob * tiles * (ncts / NCT);
      const double rounds = std::ceil((double)njobs / G);
      const double groups = (double)kmax * KQ / KS;                         // K groups per wave (the longest branch)
      const double cost = rounds * (4.0 + groups * NRT * 0.06 + WR * LDX * 4.0 / 1024.0 / 30.0);
      if (cost < best) { best = cost; found = true; *out = Geo{NRT, NCT, KS, spt, tps, tiles, ncts / NCT, (int)njobs, lds}; }
    }
  }
  return found;
}
}  // namespace

// The phase list of a vocoder step of n slots x frames frames (the rings and weights are the stream-set's; the step's mel chunk,
// audio buffers and taps arrive with the launch: VCIO)
bool conan_streams::chain_build(int n, int frames, std::vector<cnk::VCPhase>& out, int* lds_bytes, int* grid) const {
  const conan_cfg& c = ctx->cfg;
  const float LR = 0.1f;      // LRELU_SLOPE, hifigan_causal.py:20
  const int NB = c.voc_num_resblocks, ND = c.voc_rb_num_dil;
  int maxlds = 0;
  const int G = std::max(8, std::min(kChainGridMax, ctx->num_cu - 64));
  auto base_prob = [&](const PackedConv& pc, const std::string& name) {
    cnk::VCProb p; memset(&p, 0, sizeof(p));
    p.xnew[0] = p.xnew[1] = p.xnew[2] = p.xhist = p.y = p.res = ch::null_ref();
    p.w = ctx->chain_weight(name); p.bias = pc.bias;
    p.nsrc = 1; p.tap = -1; p.tap_new = -1;
    p.Cin = pc.Cin; p.Cout = pc.Cout; p.k = pc.k; p.dil = 1; p.KQ = pc.Cin / 16; p.ncts = (pc.Cout + 15) / 16;
    p.out_act = cnk::ACT_NONE; p.in_slope = 1.f; p.out_slope = 0.f; p.mean_slope = 1.f;
    p.shuffle_r = pc.shuffle_r; p.Cq = pc.Cout / std::max(1, pc.shuffle_r);
    return p;
  };
  auto push = [&](int nprob, cnk::VCProb* pr, int T) {
    cnk::VCPhase ph; memset(&ph, 0, sizeof(ph));
    // heaviest branch first: its jobs are dealt first
    std::stable_sort(pr, pr + nprob, [](const cnk::VCProb& a, const cnk::VCProb& b) { return a.k > b.k; });
    int kmax = 0, halo = 0;
    for (int q = 0; q < nprob; ++q) { kmax = std::max(kmax, pr[q].k); halo = std::max(halo, (pr[q].k - 1) * pr[q].dil); }
    Geo g;
    if (!plan_phase(n, T, pr[0].Cin, pr[0].ncts, nprob, kmax, halo, G, &g)) return false;
    ph.type = 0; ph.nprob = nprob; ph.n = n; ph.T = T; ph.NRT = g.NRT; ph.NCT = g.NCT; ph.KS = g.KS; ph.tiles_per_slot = g.tps; ph.spt = g.spt;
    ph.tiles = g.tiles; ph.ncg = g.ncg; ph.njobs = g.njobs;
    ph.magic_c4 = (int)(unsigned)((1ull << 32) / (unsigned)(pr[0].Cin / 4) + 1ull);
    for (int q = 0; q < nprob; ++q) ph.p[q] = pr[q];
    maxlds = std::max(maxlds, g.lds_floats);
    out.push_back(ph);
    return true;
  };
  out.clear();
  {  // conv_pre: the mel chunk (launch argument) + 6 frames of left context from the mel ring, to which the chunk is appended
    const PackedConv& pc = ctx->conv("voc.conv_pre");
    cnk::VCProb p = base_prob(pc, "voc.conv_pre");
    p.xnew[0] = ch::lin_ref(nullptr, frames, c.num_mels); p.io_in = 1; p.xhist = v_mel.ref(); p.store_new = 1;
    p.y = v_pre.ref(); p.out_act = cnk::ACT_LRELU; p.out_slope = LR; p.tap = 0;
    if (!push(1, &p, frames)) return false;
  }
  int ridx = 0;
  for (int i = 0; i < c.voc_num_ups; ++i) {
    const VocStage& s = v_st[i];
    const int Tin = frames * (s.rate / c.voc_up_rates[i]), T = frames * s.rate;
    {  // x = ups[i](leaky_relu(x)): the first reads conv_pre's activated output, the others form leaky_relu(mean of the branches)
      const std::string nm = "voc.ups." + std::to_string(i);
      cnk::VCProb p = base_prob(ctx->conv(nm), nm);
      if (i == 0) { p.xnew[0] = p.xhist = v_pre.ref(); }
      else {
        const VocStage& q = v_st[i - 1];
        for (int b = 0; b < NB; ++b) p.xnew[b] = q.xo[b][ND - 1].ref();
        p.nsrc = NB; p.xhist = q.xs.ref(); p.store_new = 1; p.mean_slope = LR; p.tap_new = 1 + c.voc_num_ups + (i - 1);
      }
      p.y = s.up.ref(); p.tap = 1 + i;
      if (!push(1, &p, Tin)) return false;
    }
    for (int d = 0; d < ND; ++d) {
      cnk::VCProb p1[3], p2[3];
      for (int b = 0; b < NB; ++b) {
        const std::string base = "voc.rb." + std::to_string(ridx + b);
        const TRef xin = d == 0 ? s.up.ref() : s.xo[b][d - 1].ref();
        p1[b] = base_prob(ctx->conv(base + ".c1." + std::to_string(d)), base + ".c1." + std::to_string(d));
        p1[b].xnew[0] = p1[b].xhist = xin; p1[b].in_lrelu = 1; p1[b].in_slope = LR; p1[b].dil = c.voc_rb_dilations[b][d];
        p1[b].y = s.xt[b][d].ref(); p1[b].out_act = cnk::ACT_LRELU; p1[b].out_slope = LR;
        p2[b] = base_prob(ctx->conv(base + ".c2." + std::to_string(d)), base + ".c2." + std::to_string(d));
        p2[b].xnew[0] = p2[b].xhist = s.xt[b][d].ref(); p2[b].y = s.xo[b][d].ref(); p2[b].res = xin; p2[b].has_res = 1;
      }
      if (!push(NB, p1, T) || !push(NB, p2, T)) return false;
    }
    ridx += NB;
  }
  {  // conv_post + tanh on leaky_relu(mean of the last stage's branches)
    const VocStage& s = v_st.back();
    const int T = frames * s.rate;
    cnk::VCPhase ph; memset(&ph, 0, sizeof(ph));
    ph.type = 1; ph.nprob = 1; ph.n = n; ph.T = T; ph.NRT = 1; ph.NCT = 1; ph.KS = 8;
    ph.tiles_per_slot = (T + kChainPostRows - 1) / kChainPostRows; ph.tiles = n * ph.tiles_per_slot; ph.ncg = 1; ph.njobs = ph.tiles;
    ph.magic_c4 = (int)(unsigned)((1ull << 32) / (unsigned)(s.C / 4) + 1ull);
    ph.kpost = (int)ctx->scalars.at("voc.conv_post.k");
    ph.bpost = ctx->scalars.at("voc.conv_post.b");
    ph.wpost = ctx->vec("voc.conv_post.w");
    cnk::VCProb& p = ph.p[0];
    p.xnew[0] = p.xnew[1] = p.xnew[2] = p.y = p.res = ch::null_ref();
    for (int b = 0; b < NB; ++b) p.xnew[b] = s.xo[b][ND - 1].ref();
    p.nsrc = NB; p.xhist = s.xs.ref(); p.store_new = 1; p.mean_slope = LR; p.tap = -1; p.tap_new = 1 + c.voc_num_ups + (c.voc_num_ups - 1);
    p.Cin = s.C; p.Cout = 1; p.k = ph.kpost; p.dil = 1;
    maxlds = std::max(maxlds, kChainHdrFloats + (kChainPostRows + ph.kpost - 1) * (s.C + 4) + ph.kpost * s.C);
    out.push_back(ph);
  }
  if ((int)out.size() > cnk::VC_MAX_PHASES || (size_t)maxlds * 4 > (size_t)kChainLdsMax) return false;
  *lds_bytes = maxlds * 4;
  *grid = std::min(G, cnk::voc_chain_max_grid(*lds_bytes, ctx->num_cu));
  // every tensor is addressed with 32-bit byte offsets from its base
  for (auto& ph : out)
    for (int q = 0; q < ph.nprob; ++q) {
      const TRef* ts[] = {&ph.p[q].xnew[0], &ph.p[q].xnew[1], &ph.p[q].xnew[2], &ph.p[q].xhist, &ph.p[q].y, &ph.p[q].res};
      for (const TRef* t : ts) if ((long long)max_slots * t->slot_stride * 4 >= (1ll << 31)) return false;
    }
  return *grid >= 8;
}

const conan_streams::VCProgram& conan_streams::chain_program(int n, int frames) {
  for (auto& v : vc_progs) if (v.n == n && v.frames == frames) return v;
  std::vector<cnk::VCPhase> ph;
  VCProgram v; v.n = n; v.frames = frames;
  if (!chain_build(n, frames, ph, &v.lds_bytes, &v.grid)) throw Error(CONAN_ERR_UNSUPPORTED, "voc_chain: no plan for this step (slots x frames)");
  v.nphases = (int)ph.size();
  for (auto& p : ph)
    for (int q = 0; q < p.nprob; ++q) v.flops += 2.0 * (double)n * p.T * p.p[q].Cout * p.p[q].k * p.p[q].Cin;
  HIP_CHECK(hipMalloc((void**)&v.dev, sizeof(cnk::VCPhase) * ph.size()));
  HIP_CHECK(hipMemcpy(v.dev, ph.data(), sizeof(cnk::VCPhase) * ph.size(), hipMemcpyHostToDevice));
  vc_progs.push_back(v);
  return vc_progs.back();
}

void conan_streams::chain_step(int n, int frames, const float* mel_dev, float* wav_out, float* pre_tanh, hipStream_t st, const conan_hifigan_taps* taps) {
  const conan_cfg& c = ctx->cfg;
  if ((long long)n * frames > kChainRows) throw Error(CONAN_ERR_INVALID, "more slots x frames in a step than this stream-set was created for");
  for (int i = 0; i < n; ++i) voc_fresh[h_slots[i]] = 0;
  const VCProgram& pr = chain_program(n, frames);
  cnk::VCLaunch l; memset(&l, 0, sizeof(l));
  l.prog = pr.dev; l.nphases = pr.nphases; l.grid = pr.grid; l.lds_bytes = pr.lds_bytes;
  l.slots = d_slots; l.pos = pos_voc; l.n = n; l.adv = frames; l.bar = vc_bar; l.guard = d_guard;
  l.io.mel = mel_dev; l.io.wav = wav_out; l.io.pre = pre_tanh;
  if (taps) {
    l.io.tap[0] = taps->conv_pre_act;
    for (int i = 0; i < c.voc_num_ups; ++i) { l.io.tap[1 + i] = taps->ups[i]; l.io.tap[1 + c.voc_num_ups + i] = taps->stage_out[i]; }
  }
  static const bool stamps = getenv("CONAN_VC_STAMPS") != nullptr;
  static unsigned long long* dbg = nullptr;
  const size_t dbg_words = (size_t)kChainGridMax * cnk::VC_MAX_PHASES * 4 + kChainGridMax + 1;
  if (stamps) {
    if (!dbg) HIP_CHECK(hipMalloc((void**)&dbg, dbg_words * 8));
    HIP_CHECK(hipMemsetAsync(dbg, 0, dbg_words * 8, st));
    l.dbg = dbg;
  }
  profiled("cnk::voc_chain_kernel", pr.flops, st, [&] {
    // one chain launch of this context at a time on the device: each needs its whole grid resident
    std::lock_guard<std::mutex> lock(ctx->chain_mu);
    if (!ctx->chain_done) HIP_CHECK(hipEventCreateWithFlags(&ctx->chain_done, hipEventDisableTiming));
    else HIP_CHECK(hipStreamWaitEvent(st, ctx->chain_done, 0));
    cnk::launch_voc_chain(l, st);
    HIP_CHECK(hipEventRecord(ctx->chain_done, st));
  });
  if (stamps) {
    static int calls = 0;
    if (++calls == 40) {      // (a warm launch)
      HIP_CHECK(hipStreamSynchronize(st));
      std::vector<unsigned long long> h(dbg_words);
      HIP_CHECK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
      std::vector<cnk::VCPhase> ph(pr.nphases);
      HIP_CHECK(hipMemcpy(ph.data(), pr.dev, sizeof(cnk::VCPhase) * pr.nphases, hipMemcpyDeviceToHost));
      const int G = pr.grid;
      const size_t tail = (size_t)G * cnk::VC_MAX_PHASES * 4;
      unsigned long long t0 = ~0ull;
      for (int b = 0; b < G; ++b) t0 = std::min(t0, h[tail + b]);
      fprintf(stderr, "[voc_chain] n %d frames %d grid %d lds %d B: %.2f us in all (first workgroup's start -> the last one's end)\n", n, frames, G, pr.lds_bytes, (h[tail + G] - t0) / 100.0);
      double prev = 0.0;
      for (int p = 0; p < pr.nphases; ++p) {
        double v[4] = {0, 0, 0, 0}, first_wait = 1e30;
        for (int b = 0; b < G; ++b)
          for (int q = 0; q < 4; ++q) {
            const unsigned long long x = h[((size_t)b * cnk::VC_MAX_PHASES + p) * 4 + q];
            if (x) { v[q] = std::max(v[q], (x - t0) / 100.0); if (q == 0) first_wait = std::min(first_wait, (x - t0) / 100.0); }
          }
        fprintf(stderr, "  phase %2d type %d T %5d Cin %3d Cout %4d k %2d nprob %d NRT %d NCT %d KS %d jobs %4d: wait passed %7.2f .. %7.2f gathered %7.2f K loops %7.2f stores issued %7.2f  (+%.2f)\n", p, ph[p].type, ph[p].T,
                ph[p].p[0].Cin, ph[p].p[0].Cout, ph[p].p[0].k, ph[p].nprob, ph[p].NRT, ph[p].NCT, ph[p].KS, ph[p].njobs, first_wait, v[0], v[1], v[2], v[3], v[3] - prev);
        if (v[3] > 0) prev = v[3];
      }
    }
  }
}
