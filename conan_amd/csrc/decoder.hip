// Launch plans of the Conan decoder step (content-dependent part of Conan.forward, infer=True) and of
// the per-utterance style pass (reference-mel-only part), see modules/Conan/Conan.py:115-270,:324-351,:584-589.
#include "streams.h"

static cnk::LNArgs mk_ln(const TRef& x, const TRef& y, float* g, float* b, const int* slots, const int* pos, int n, int T, int C) {
  cnk::LNArgs a; memset(&a, 0, sizeof(a));
  a.x = x; a.y = y; a.gamma = g; a.beta = b; a.slots = slots; a.pos = pos; a.lens = nullptr; a.T = T; a.n = n; a.C = C; a.eps = 1e-5f;
  return a;
}

// ------------------------------------------------------------------------------------------------ operator sinks
// Every launch of the decoder step goes through one of these (and conan_streams::rowconv / launch_group): they either
// launch, or - while run_mega() records the step - append the operator to the megakernel's program.

void conan_streams::mega_push(cnk::MegaOp& op, int lds_floats) {
  op.barrier = 1;
  mega_rec_lds = std::max(mega_rec_lds, lds_floats);
  if ((int)mega_rec->size() >= kMegaMaxOps) { mega_rec_ok = false; return; }
  mega_rec->push_back(op);
}
void conan_streams::op_embed(const cnk::EmbedArgs& a, hipStream_t st) {
  if (!mega_rec) { cnk::launch_embed(a, st); return; }
  cnk::MegaOp op; memset(&op, 0, sizeof(op));
  op.type = cnk::MOP_EMBED; op.nbx = a.n * a.T; op.nby = 1; op.u.em = a;
  mega_push(op, 0);
}
void conan_streams::op_ln(const cnk::LNArgs& a, hipStream_t st) {
  if (!mega_rec) { cnk::launch_layernorm(a, st); return; }
  if (a.C > 512) { mega_rec_ok = false; return; }
  cnk::MegaOp op; memset(&op, 0, sizeof(op));
  op.type = cnk::MOP_LN; op.nbx = (a.n * a.T + 3) / 4; op.nby = 1; op.u.ln = a;
  mega_push(op, 0);
}
void conan_streams::op_xattn(const cnk::XAttnArgs& a, hipStream_t st) {
  if (!mega_rec) { cnk::launch_xattn(a, st); return; }
  if (a.H > 4 || a.E > 1024 || a.attn_avg) { mega_rec_ok = false; return; }
  cnk::MegaOp op; memset(&op, 0, sizeof(op));
  op.type = cnk::MOP_XATTN; op.nbx = a.n * a.T; op.nby = 1; op.u.xa = a;
  cnk::MegaOp probe = op;
  mega_push(op, cnk::decoder_mega_lds_floats(probe, 0));
}
void conan_streams::op_pitch(const cnk::PitchHeadArgs& a, hipStream_t st) {
  if (!mega_rec) { cnk::launch_pitch_head(a, st); return; }
  cnk::MegaOp op; memset(&op, 0, sizeof(op));
  op.type = cnk::MOP_PITCH; op.nbx = (a.n * a.T + 3) / 4; op.nby = 1; op.u.ph = a;
  const double mn = 1127.0 * log(1.0 + 50.0 / 700.0), mxx = 1127.0 * log(1.0 + 900.0 / 700.0);     // launch_pitch_head's constants
  op.f0 = (float)mn; op.f1 = (float)(mxx - mn);
  mega_push(op, 0);
}
void conan_streams::op_advance(int* pos, int n, int delta, hipStream_t st) {
  if (!mega_rec) { cnk::launch_advance(pos, d_slots, n, delta, st); return; }
  cnk::MegaOp op; memset(&op, 0, sizeof(op));
  op.type = cnk::MOP_ADVANCE; op.nbx = 1; op.nby = 1;
  op.u.adv.pos = pos; op.u.adv.slots = d_slots; op.u.adv.n = n; op.u.adv.delta = delta;
  mega_push(op, 0);
  mega_rec->back().barrier = 0;       // the step's last operator
}

// The decoder step as ONE persistent launch.  The operator list of a (slot count, frames, buffer set) is recorded once by
// running decoder_ops() in recording mode, kept in a small LRU cache (the pipelined step rotates through 4 hand-off
// buffers: 4 entries that hit for ever) and replayed.  false: this step has an operator the megakernel does not cover.
bool conan_streams::run_mega(int n, int T, const int32_t* codes, float* mel_out, const DecExtra& ex, hipStream_t st) {
  const long long key[6] = {((long long)n << 32) | (unsigned)T, (long long)(uintptr_t)codes, (long long)(uintptr_t)mel_out, (long long)(uintptr_t)ex.mel_out2,
                            (long long)(uintptr_t)ex.codes_dst, (long long)(uintptr_t)ex.codes_src ^ ((long long)ex.codes_words << 48)};
  MegaProgram* e = nullptr;
  for (auto& m : mega_cache) if (m.dev && memcmp(m.key, key, sizeof(key)) == 0) { e = &m; break; }
  if (!e) {
    if ((int)mega_cache.size() < kMegaEntries) { mega_cache.emplace_back(); e = &mega_cache.back(); }
    else { e = &mega_cache[0]; for (auto& m : mega_cache) if (m.stamp < e->stamp) e = &m; }
    if (!e->dev) {
      HIP_CHECK(hipMalloc((void**)&e->dev, sizeof(cnk::MegaOp) * kMegaMaxOps));
      HIP_CHECK(hipHostMalloc((void**)&e->pinned, sizeof(cnk::MegaOp) * kMegaMaxOps, hipHostMallocDefault));
      HIP_CHECK(hipEventCreateWithFlags(&e->copied, hipEventDisableTiming));
    }
    std::vector<cnk::MegaOp> ops;
    mega_rec = &ops; mega_rec_ok = true; mega_rec_lds = 0; mega_rec_flops = 0.0;
    // members a fused feed-forward's hidden columns are split over: the group's workgroups, or - a single row tile, whose group forms
    // at run time (xcd mode) - one virtual member per 64 hidden columns
    mega_ffn_gs = plan_n(n) * T <= 16 ? std::max(1, (int)ctx->conv("conan.align.0.ff1").Cout / 64) : mega_gs;
    try {
      if (ex.codes_dst) {      // the caller's copy of the step's codes: independent of everything else
        cnk::MegaOp op; memset(&op, 0, sizeof(op));
        op.type = cnk::MOP_COPY32; op.nbx = 1; op.nby = 1;
        op.u.cp.dst = reinterpret_cast<unsigned*>(ex.codes_dst); op.u.cp.src = reinterpret_cast<const unsigned*>(ex.codes_src); op.u.cp.n = ex.codes_words;
        mega_push(op, 0);
        ops.back().barrier = 0;
      }
      conan_decoder_taps none; memset(&none, 0, sizeof(none));
      decoder_ops(n, T, codes, mel_out, none, st);
      if (ex.mel_out2 && mega_rec_ok) {
        // the caller's copy of the mel frames: the mel_out layer once more into the second buffer (the operator before the
        // counter advance is mel_out; both read the same input, no barrier between them)
        const int lt = ops.size() >= 2 ? ops[ops.size() - 2].type : -1;
        if (ops.back().type != cnk::MOP_ADVANCE || (lt != cnk::MOP_RC111 && lt != cnk::MOP_RC114)) mega_rec_ok = false;
        else {
          cnk::MegaOp twin = ops[ops.size() - 2];
          twin.u.rc.y = ch::lin_ref(ex.mel_out2, T, ctx->cfg.num_mels);
          ops[ops.size() - 2].barrier = 0;
          ops.insert(ops.end() - 1, twin);
        }
      }
    } catch (...) { mega_rec = nullptr; throw; }
    mega_rec = nullptr;
    memcpy(e->key, key, sizeof(key));
    e->ok = mega_rec_ok && !ops.empty() && (int)ops.size() <= kMegaMaxOps;
    if (e->ok) {
      // geometry: njobs 16-row tiles; one group per tile (at most mega_grid / 8 groups) of GS workgroups; a single tile is
      // worked on by the whole grid as one group, with 16-column strips (K split over the waves of a workgroup)
      int nb = 0, maxs = 1;
      for (auto& op : ops) {
        nb += op.barrier ? 1 : 0;
        if (op.type <= cnk::MOP_ROWLIN) maxs = std::max(maxs, op.nbx);
      }
      const int njobs = (n * T + 15) / 16;
      e->njobs = njobs; e->kw4 = 0; e->n = n; e->T = T;
      // A single tile (<= 16 rows: one to four streams): xcd mode - one workgroup per CU is launched (enough dynamic LDS that two do
      // not share a CU), the ~32 that land on workgroup 0's XCD walk the program as ONE group whose hand-offs stay in that XCD's L2
      // (plain stores, L1-bypassing loads, flag barriers: ~1 us per operator instead of ~5 through memory), the others leave at once.
      e->xcd = plan_n(n) * T <= 16;      // (fixed-plan stream-sets: by max_slots - a single active tile of a larger set runs the multi-tile form)
      if (e->xcd) { e->groups = 1; e->group_size = ctx->num_cu; }
      else {
        e->group_size = mega_gs; e->groups = std::max(1, std::min(njobs, mega_grid / mega_gs));
        // (group-fastest layout: with a group count that is a multiple of 8 a group's members share an XCD - decoder_mega.hip, CM = 3;
        // groups beyond the job count have no tile and only join the launch's last barrier)
        const int padded = (e->groups + 7) / 8 * 8;
        if (padded * mega_gs <= std::max(mega_grid, 64) && padded <= 32) e->groups = padded;
      }
      e->nops = (int)ops.size(); e->barriers = nb; e->flops = mega_rec_flops;
      // (xcd mode, blocking steps: enough dynamic LDS that two workgroups do not share a CU - one per CU, ~32 on the elected XCD;
      // pipelined steps launch with what the operators need, so that a vocoder workgroup fits beside a member: launch_mega)
      e->lds_need = mega_rec_lds * 4;
      e->lds_bytes = e->xcd ? std::max(mega_rec_lds * 4, 84 * 1024) : mega_rec_lds * 4;
      // the arrival-counter barriers need every workgroup of the grid resident at once: never launch more than the device can hold
      // (a quarter of the CUs is kept as margin for what else is resident); such a step keeps its separate launches
      const long long cap = (long long)cnk::decoder_mega_blocks_per_cu(e->lds_bytes, rb_limb) * (ctx->num_cu - ctx->num_cu / 4);
      // (xcd mode: only the elected XCD's workgroups stay, the others leave at once - nothing waits for the whole grid to be resident)
      if (!e->xcd && (long long)e->groups * e->group_size > cap) e->ok = false;
      if (e->xcd && e->lds_bytes > 126 * 1024) e->ok = false;
    }
    if (e->ok) {
      HIP_CHECK(hipEventSynchronize(e->copied));            // (the entry's previous upload, if any, has long completed)
      memcpy(e->pinned, ops.data(), sizeof(cnk::MegaOp) * ops.size());
      HIP_CHECK(hipMemcpyAsync(e->dev, e->pinned, sizeof(cnk::MegaOp) * ops.size(), hipMemcpyHostToDevice, st));
      HIP_CHECK(hipEventRecord(e->copied, st));
    }
  }
  e->stamp = ++mega_clock;
  if (!e->ok) return false;
  launch_mega(*e, st);
  return true;
}

void conan_streams::decoder_step(int n, int T, const int32_t* codes, float* mel_out, const conan_decoder_taps& taps, hipStream_t st, const DecExtra* extra) {
  for (int i = 0; i < n; ++i)
    if (!has_ref[h_slots[i]]) throw Error(CONAN_ERR_STATE, "conan_decoder_step before conan_set_reference for slot " + std::to_string(h_slots[i]) +
                                                            " (the reference raises ValueError when ref is None)");
  const bool notaps = !taps.uv_pred && !taps.f0_denorm_pred && !taps.pitch_bins && !taps.decoder_inp && !taps.content_embed_proj && !taps.attn[0] && !taps.attn[1];
  DecExtra ex; if (extra) ex = *extra;
  // (a job is a 16-row tile taken through the whole operator list by its own group of workgroups: a stream's frames of the
  // step must all fall into one tile - 16 % frames == 0, or a single tile in all)
  // A single tile (<= 4 streams): round 3 kept the separate launches for it - a grid-wide barrier through memory per operator cost
  // what the launch boundaries do (0.47 against 0.39 ms at one stream).  Round 5: such steps run the persistent launch in xcd mode
  // (run_mega), whose barriers and hand-offs stay inside one XCD's L2; mega_single (conan_streams_opts / CONAN_MEGA_SINGLE=0) turns it off.
  const int prows = plan_n(n) * T;      // the rows the plan is made for: the step's own, or - CONAN_STREAMS_FIXED_PLAN - the full stream-set's
  const bool tiles_ok = (16 % T == 0 && prows > 16) || (prows <= 16 && T >= 2 && mega_single);
  // (the per-op Emformer plan - memory bank, shapes the fused step does not cover - is ~90 launches whose conv_mfma workgroups
  // need CUs of their own: beside 128 resident decoder workgroups they queue, b128s2mem4 2.13 against 2.00 ms per step)
  const bool emf_ok = !(ctx->cfg.models & CONAN_MODEL_EMFORMER) || emf_fused;
  if (use_mega && notaps && tiles_ok && emf_ok && mega_bar && run_mega(n, T, codes, mel_out, ex, st)) return;
  if (ex.codes_dst) HIP_CHECK(hipMemcpyAsync(ex.codes_dst, ex.codes_src, (size_t)ex.codes_words * sizeof(int), hipMemcpyDeviceToDevice, st));
  decoder_ops(n, T, codes, mel_out, taps, st);
  if (ex.mel_out2) HIP_CHECK(hipMemcpyAsync(ex.mel_out2, mel_out, (size_t)n * T * ctx->cfg.num_mels * sizeof(float), hipMemcpyDeviceToDevice, st));
}

void conan_streams::decoder_ops(int n, int T, const int32_t* codes, float* mel_out, const conan_decoder_taps& taps, hipStream_t st) {
  float* const uv_pred = taps.uv_pred; float* const f0 = taps.f0_denorm_pred; int32_t* const bins = taps.pitch_bins;
  float* const dec_inp = taps.decoder_inp;
  const conan_cfg& c = ctx->cfg;
  const int H = c.hidden_size;
  const int* pos = pos_dec;
  // content_embedding (Conan.py:140)
  {
    cnk::EmbedArgs a; memset(&a, 0, sizeof(a));
    a.y = c_emb.ref(); a.table = ctx->vec("conan.content_embedding"); a.idx = codes; a.slots = d_slots; a.pos = pos;
    a.T = T; a.n = n; a.C = H; a.vocab = c.content_vocab;
    op_embed(a, st);
  }
  // content_proj: CausalConv1d k3 + LeakyReLU(0.01) (Conan.py:57-60, :142); pitch_inp = content + style (Conan.py:162)
  if (rowconv_ok(ctx->conv("conan.content_proj"), 1, T)) {
    cnk::RowConvArgs a = mk_rc(ctx->conv("conan.content_proj"), c_emb.ref(), c_pin.ref(), n, T);
    a.out_act = cnk::ACT_LRELU; a.out_slope = 0.01f; a.bvec = c_style; a.bvec_stride = H;
    rowconv(a, st);
    if (taps.content_embed_proj) {
      cnk::RowConvArgs t = mk_rc(ctx->conv("conan.content_proj"), c_emb.ref(), ch::lin_ref(taps.content_embed_proj, T, H), n, T);
      t.out_act = cnk::ACT_LRELU; t.out_slope = 0.01f;
      rowconv(t, st);
    }
  } else {
    ConvArgs a = mk(ctx->conv("conan.content_proj"), c_emb.ref(), c_pin.ref(), n, T, pos);
    a.out_act = cnk::ACT_LRELU; a.out_slope = 0.01f; a.bvec = c_style; a.bvec_stride = H;
    conv(a, st);
    if (taps.content_embed_proj) {   // the tap is the same conv without the fused "+ style" (exact, tap mode only)
      ConvArgs t = mk(ctx->conv("conan.content_proj"), c_emb.ref(), ch::lin_ref(taps.content_embed_proj, T, H), n, T, pos);
      t.out_act = cnk::ACT_LRELU; t.out_slope = 0.01f;
      conv(t, st);
    }
  }
  // ProsodyAligner: 2 x CrossAttenLayer, post-LN (prosody_util.py:119-126)
  const int nh = 2, dh = H / nh;
  Lin* src = &c_pin;
  // Post-LN layers: where the consumer of a LayerNorm is a rowconv launch the norm is that launch's prologue (the
  // normalised rows also go to `hist`, where the residual adds read them): norm1 -> ff1, and layer 0's norm2 -> layer 1's q.
  bool n2_pending = false;                       // the previous layer's norm2 is still to be applied (by this layer's q)
  // Megakernel, several row tiles: the feed-forward of a layer is ONE operator (ff1 -> ReLU -> ff2 with the hidden columns
  // split over the 8 members of a group, decoder_mega.hip MOP_FFN); what ff2 would have written to c_a1 then exists as 8
  // partial tensors + bias + residual, summed by whoever reads it (the norm2 behind it).
  static const bool ffn_fuse_on = ch::dev_getenv("CONAN_MEGA_NOFFN") == nullptr;
  bool ffn_parts = false;
  const long long part_stride = (long long)max_slots * max_frames * H;
  auto x_parts = [&](auto& a, const PackedConv& ff2) {
    a.xp = c_part.base; a.xp_stride = part_stride; a.xparts = mega_ffn_gs; a.xp_ld = H; a.xbias = ff2.bias; a.xres = c_a2.ref(); a.has_xres = 1;
  };
  for (int l = 0; l < 2; ++l) {
    const std::string nm = "conan.align." + std::to_string(l);
    if (rowconv_ok(ctx->conv(nm + ".q"), 1, T)) {
      cnk::RowConvArgs a = mk_rc(ctx->conv(nm + ".q"), n2_pending ? c_a1.ref() : src->ref(), c_q.ref(), n, T);
      if (n2_pending) { a.ln = 1; a.hist = src->ref(); a.gamma = ctx->vec("conan.align." + std::to_string(l - 1) + ".norm2.g"); a.beta = ctx->vec("conan.align." + std::to_string(l - 1) + ".norm2.b"); }
      if (n2_pending && ffn_parts) x_parts(a, ctx->conv("conan.align." + std::to_string(l - 1) + ".ff2"));
      a.out_scale = (float)std::sqrt(1.0 / (double)dh); rowconv(a, st);
    } else { ConvArgs a = mk(ctx->conv(nm + ".q"), src->ref(), c_q.ref(), n, T, pos); a.out_scale = (float)std::sqrt(1.0 / (double)dh); conv(a, st); }
    n2_pending = false;
    {
      cnk::XAttnArgs a; memset(&a, 0, sizeof(a));
      a.q = c_q.ref(); a.out = c_att.ref(); a.kv = c_kv + (size_t)l * S_max * 2 * H; a.kv_slot_stride = (long long)2 * S_max * 2 * H;
      a.kmask = c_kmask; a.slen = c_slen; a.attn_avg = taps.attn[l]; a.slots = d_slots; a.pos = pos; a.T = T; a.n = n; a.E = H; a.H = nh; a.S_max = S_max;
      op_xattn(a, st);
    }
    if (rowconv_ok(ctx->conv(nm + ".out"), 1, T)) { cnk::RowConvArgs a = mk_rc(ctx->conv(nm + ".out"), c_att.ref(), c_a1.ref(), n, T); a.res = src->ref(); a.has_res = 1; rowconv(a, st); }
    else { ConvArgs a = mk(ctx->conv(nm + ".out"), c_att.ref(), c_a1.ref(), n, T, pos); a.res = src->ref(); a.has_res = 1; conv(a, st); }
    ffn_parts = false;
    {
      const PackedConv &f1 = ctx->conv(nm + ".ff1"), &f2 = ctx->conv(nm + ".ff2");
      // (a single row tile - xcd mode, decoder_mega.hip - splits the hidden columns over Cout / 64 virtual members: mega_ffn_gs)
      if (mega_rec && ffn_fuse_on && rowconv_ok(f1, 1, T) && f2.wf && f1.k == 1 && f2.k == 1 && f1.Cout % (64 * mega_ffn_gs) == 0 && f1.Cout / mega_ffn_gs <= 256 &&
          (plan_n(n) * T > 16 || f1.Cout / mega_ffn_gs == 64) && mega_ffn_gs <= 32 && f2.Cin == f1.Cout && f2.Cout == H && H % 64 == 0) {
        cnk::RowConvArgs a = mk_rc(f1, c_a1.ref(), c_ff.ref(), n, T);
        a.ln = 1; a.hist = c_a2.ref(); a.gamma = ctx->vec(nm + ".norm1.g"); a.beta = ctx->vec(nm + ".norm1.b");
        a.out_act = cnk::ACT_RELU;
        a.w2 = f2.wf; a.part = c_part.base; a.part_stride = part_stride; a.Cout2 = f2.Cout; a.Cout2_pad = f2.wf_cout_pad;
        rowconv(a, st);
        ffn_parts = true;
      }
    }
    if (ffn_parts) {
    } else if (rowconv_ok(ctx->conv(nm + ".ff1"), 1, T)) {       // norm1 (prologue; normalised rows -> c_a2, ff2's residual) -> ff1 -> ReLU
      cnk::RowConvArgs a = mk_rc(ctx->conv(nm + ".ff1"), c_a1.ref(), c_ff.ref(), n, T);
      a.ln = 1; a.hist = c_a2.ref(); a.gamma = ctx->vec(nm + ".norm1.g"); a.beta = ctx->vec(nm + ".norm1.b");
      a.out_act = cnk::ACT_RELU; rowconv(a, st);
    } else {
      op_ln(mk_ln(c_a1.ref(), c_a2.ref(), ctx->vec(nm + ".norm1.g"), ctx->vec(nm + ".norm1.b"), d_slots, pos, n, T, H), st);
      ConvArgs a = mk(ctx->conv(nm + ".ff1"), c_a2.ref(), c_ff.ref(), n, T, pos); a.out_act = cnk::ACT_RELU; conv(a, st);
    }
    // ff2 (K = 2048) -> c_a1 (free again): rowlin - a 33 KB block that shares CUs with the vocoder's, where the split-K conv_mfma
    // build (126 KB of LDS) needs CUs of its own; a handful of rows (one row tile) keep the split-K build, which spreads K over blocks
    if (ffn_parts) {
    } else if (rowconv_ok(ctx->conv(nm + ".ff2"), 1, T) && (plan_n(n) * T > 16 || mega_rec)) { cnk::RowConvArgs a = mk_rc(ctx->conv(nm + ".ff2"), c_ff.ref(), c_a1.ref(), n, T); a.res = c_a2.ref(); a.has_res = 1; rowconv(a, st); }
    else { ConvArgs a = mk(ctx->conv(nm + ".ff2"), c_ff.ref(), c_a1.ref(), n, T, pos); a.res = c_a2.ref(); a.has_res = 1; conv(a, st); }
    if (l == 0 && rowconv_ok(ctx->conv("conan.align.1.q"), 1, T)) n2_pending = true;      // norm2 -> c_x[0] happens in layer 1's q launch
    else {
      cnk::LNArgs ln = mk_ln(c_a1.ref(), l == 0 ? c_x[0].ref() : c_pin2.ref(), ctx->vec(nm + ".norm2.g"), ctx->vec(nm + ".norm2.b"), d_slots, pos, n, T, H);
      if (l == 1) { ln.post = c_pin.ref(); ln.has_post = 1; }   // pitch_inp = pitch_inp + prosody (Conan.py:168)
      if (ffn_parts) x_parts(ln, ctx->conv(nm + ".ff2"));
      op_ln(ln, st);
    }
    src = &c_x[0];
  }
  // uv_predictor: 5 x [CausalConv k5 + ReLU] (nar_tts_modules.py:113-122)
  const int n_uv = (int)ctx->scalars.at("conan.uv.n");
  for (int i = 0; i < n_uv; ++i) {
    const TRef xin = i == 0 ? c_pin2.ref() : c_uvh[i - 1].ref();
    const TRef yout = i == n_uv - 1 ? c_uv5.ref() : c_uvh[i].ref();
    const PackedConv& pc = ctx->conv("conan.uv." + std::to_string(i));
    if (rowconv_ok(pc, 1, T)) { cnk::RowConvArgs a = mk_rc(pc, xin, yout, n, T); a.out_act = cnk::ACT_RELU; rowconv(a, st); continue; }
    ConvArgs a = mk(pc, xin, yout, n, T, pos);
    a.out_act = cnk::ACT_RELU;
    conv(a, st);
  }
  {
    cnk::PitchHeadArgs a; memset(&a, 0, sizeof(a));
    a.h = c_uv5.ref(); a.pitch_inp = c_pin2.ref(); a.dec_inp = c_x[0].ref();
    a.gamma = ctx->vec("conan.uv.ln.g"); a.beta = ctx->vec("conan.uv.ln.b"); a.w = ctx->vec("conan.uv.lin.w"); a.b = ctx->vec("conan.uv.lin.b");
    a.pitch_embed = ctx->vec("conan.pitch_embed"); a.codes = codes; a.uv_pred = uv_pred; a.f0 = f0; a.bins = bins;
    a.slots = d_slots; a.pos = pos; a.T = T; a.n = n; a.Cp = (int)ctx->scalars.at("conan.uv.hidden"); a.E = H; a.silent_token = c.silent_token;
    op_pitch(a, st);
  }
  if (dec_inp) {
    cnk::CopyArgs ca; memset(&ca, 0, sizeof(ca));
    ca.n = n; ca.C = H; ca.T = T; ca.x = c_x[0].ref(); ca.y = ch::lin_ref(dec_inp, T, H);
    cnk::launch_copy_rows(ca, st);
  }
  // decoder: CausalConvBlocks (conv.py:127-264)
  int cur = 0;
  const float kscale = (float)std::pow((double)c.dec_kernel, -0.5);
  // Megakernel, several row tiles: a sub-layer [LN -> k5 conv -> GELU] -> [1x1 conv + residual, masks] is ONE operator (MOP_FFN:
  // member s of a group computes its 64 hidden columns into LDS and multiplies them at once with its K range of the 1x1 conv;
  // the 16 x 512 hidden tile never leaves the CUs, one group barrier / gather / write-through round per sub-layer is gone).  The
  // sub-layer's output x' = ((p0 + .. + p7) + bias + x) * masks then exists as 8 partial tensors until its consumer - the next
  // sub-layer's LayerNorm prologue, or the post conv's - forms it; that consumer also stores x' (its own consumer's residual).
  // Two sets of partial tensors alternate: members still read one while the faster ones already write the next.
  // MEASURED (64 streams, one box, alternating runs): the decoder step alone 0.535 against 0.542 ms and the blocking step's p50
  // 2.012 against 2.048 ms - but the pipelined step 1.440 against 1.420 ms: every member reads all 8 partial tensors (1 MB per
  // job and sub-layer through sc1 loads: +100 MB of HBM fetches per step, 253 against 150 MB for the launch), which the vocoder's
  // kernels feel.  Throughput is the headline, so the fused form is OFF unless CONAN_MEGA_BLK=1 (developer switch, read per
  // recording; tests/test_gpu_round4.py runs it against the separate launches).
  const bool blk_fuse_on = (opt_flags & CONAN_STREAMS_FUSED_DECODER_BLOCKS) != 0;
  struct BlkParts { bool on = false; const float* xp = nullptr; const float* bias = nullptr; TRef xres, m1, m2; int has_m2 = 0; } bp;
  int pset = 0;
  auto blk_consume = [&](cnk::RowConvArgs& a, bool store) {
    a.xp = bp.xp; a.xp_stride = part_stride; a.xparts = mega_gs; a.xp_ld = H; a.xbias = bp.bias; a.xres = bp.xres; a.has_xres = 1;
    a.xm1 = bp.m1; a.has_xm1 = 1; a.xm2 = bp.m2; a.has_xm2 = bp.has_m2; a.xstore = store ? 1 : 0;
  };
  for (int b = 0; b < c.dec_num_blocks; ++b) {
    // (block masks alternate between two buffers: the consumer of block b's last sub-layer reads block b's mask while it writes block b + 1's)
    const TRef blkmask = b == 0 ? c_mask_out.ref() : ((b & 1) ? c_mask_blk.ref() : c_mask_blk2.ref());
    for (int j = 0; j < c.dec_layers_in_block; ++j) {
      const std::string nm = "conan.dec." + std::to_string(b) + "." + std::to_string(j);
      Ring& lr = c_lnrs[b * c.dec_layers_in_block + j];
      const bool last_sub = b == c.dec_num_blocks - 1 && j == c.dec_layers_in_block - 1;
      {
        const PackedConv &p1 = ctx->conv(nm + ".c1"), &p2 = ctx->conv(nm + ".c2");
        if (mega_rec && blk_fuse_on && plan_n(n) * T > 16 && rowconv_ok(p1, c.dec_dilations[b], T) && p2.wf && p2.k == 1 && p1.Cout % (64 * mega_gs) == 0 &&
            p1.Cout / mega_gs <= 256 && (p1.Cout / mega_gs == 64 || (p1.Cout / mega_gs) % 128 == 0) && p2.Cin == p1.Cout && p2.Cout == H && H % 64 == 0) {
          cnk::RowConvArgs a = mk_rc(p1, c_x[cur].ref(), c_h.ref(), n, T, c.dec_dilations[b]);
          a.ln = 1; a.hist = lr.ref(); a.gamma = ctx->vec(nm + ".ln.g"); a.beta = ctx->vec(nm + ".ln.b");
          if (j == 0) { a.mask_out = blkmask; a.has_mask_out = 1; }
          a.out_scale = kscale; a.out_act = cnk::ACT_GELU;
          float* const pw = (pset ? c_part2 : c_part).base;
          a.w2 = p2.wf; a.part = pw; a.part_stride = part_stride; a.Cout2 = p2.Cout; a.Cout2_pad = p2.wf_cout_pad;
          if (bp.on) blk_consume(a, true);          // x = the previous sub-layer's output, still in parts: formed (and stored to c_x[cur]) by this operator's prologue
          rowconv(a, st);
          bp.on = true; bp.xp = pw; bp.bias = p2.bias; bp.xres = c_x[cur].ref(); bp.m1 = blkmask; bp.m2 = c_mask_out.ref(); bp.has_m2 = last_sub ? 1 : 0;
          pset ^= 1; cur ^= 1;
          continue;
        }
      }
      if (bp.on) { mega_rec_ok = false; return; }      // (a fused sub-layer followed by one that is not: this step keeps its separate launches)
      if (rowconv_ok(ctx->conv(nm + ".c1"), c.dec_dilations[b], T) && rowconv_ok(ctx->conv(nm + ".c2"), 1, T)) {
        // LayerNorm (prologue: new rows normalised in LDS and appended to the layer's ring) -> k5 conv -> x k^-0.5 -> GELU
        cnk::RowConvArgs a = mk_rc(ctx->conv(nm + ".c1"), c_x[cur].ref(), c_h.ref(), n, T, c.dec_dilations[b]);
        a.ln = 1; a.hist = lr.ref(); a.gamma = ctx->vec(nm + ".ln.g"); a.beta = ctx->vec(nm + ".ln.b");
        if (j == 0) { a.mask_out = blkmask; a.has_mask_out = 1; }
        a.out_scale = kscale; a.out_act = cnk::ACT_GELU;
        rowconv(a, st);
        cnk::RowConvArgs a2 = mk_rc(ctx->conv(nm + ".c2"), c_h.ref(), c_x[cur ^ 1].ref(), n, T);
        a2.res = c_x[cur].ref(); a2.has_res = 1; a2.m1 = blkmask; a2.has_m1 = 1;
        if (last_sub) { a2.m2 = c_mask_out.ref(); a2.has_m2 = 1; }
        rowconv(a2, st);
        cur ^= 1;
        continue;
      }
      cnk::LNArgs ln = mk_ln(c_x[cur].ref(), lr.ref(), ctx->vec(nm + ".ln.g"), ctx->vec(nm + ".ln.b"), d_slots, pos, n, T, H);
      if (j == 0) { ln.mask_out = blkmask; ln.has_mask_out = 1; }
      op_ln(ln, st);
      { ConvArgs a = mk(ctx->conv(nm + ".c1"), lr.ref(), c_h.ref(), n, T, pos, c.dec_dilations[b]); a.out_scale = kscale; a.out_act = cnk::ACT_GELU; conv(a, st); }
      {
        ConvArgs a = mk(ctx->conv(nm + ".c2"), c_h.ref(), c_x[cur ^ 1].ref(), n, T, pos);
        a.res = c_x[cur].ref(); a.has_res = 1; a.m1 = blkmask; a.has_m1 = 1;
        if (last_sub) { a.m2 = c_mask_out.ref(); a.has_m2 = 1; }
        conv(a, st);
      }
      cur ^= 1;
    }
  }
  if (bp.on && !rowconv_ok(ctx->conv("conan.dec.post"), 1, T)) { mega_rec_ok = false; return; }
  if (rowconv_ok(ctx->conv("conan.dec.post"), 1, T)) {   // last LayerNorm (x mask) as the prologue of the post conv
    cnk::RowConvArgs a = mk_rc(ctx->conv("conan.dec.post"), c_x[cur].ref(), c_post.ref(), n, T);
    a.ln = 1; a.hist = c_lastr.ref(); a.gamma = ctx->vec("conan.dec.last.g"); a.beta = ctx->vec("conan.dec.last.b");
    a.lnmask = c_mask_out.ref(); a.has_lnmask = 1; a.m1 = c_mask_out.ref(); a.has_m1 = 1;
    if (bp.on) blk_consume(a, false);               // the last sub-layer's output, from its partial tensors
    rowconv(a, st);
  } else {
    cnk::LNArgs ln = mk_ln(c_x[cur].ref(), c_lastr.ref(), ctx->vec("conan.dec.last.g"), ctx->vec("conan.dec.last.b"), d_slots, pos, n, T, H);
    ln.m1 = c_mask_out.ref(); ln.has_m1 = 1;
    op_ln(ln, st);
    ConvArgs a = mk(ctx->conv("conan.dec.post"), c_lastr.ref(), c_post.ref(), n, T, pos); a.m1 = c_mask_out.ref(); a.has_m1 = 1; conv(a, st);
  }
  if (rowconv_ok(ctx->conv("conan.mel_out"), 1, T)) rowconv(mk_rc(ctx->conv("conan.mel_out"), c_post.ref(), ch::lin_ref(mel_out, T, c.num_mels), n, T), st);
  else conv(mk(ctx->conv("conan.mel_out"), c_post.ref(), ch::lin_ref(mel_out, T, c.num_mels), n, T, pos), st);
  op_advance(pos_dec, n, T, st);
}

// ------------------------------------------------------------------------------------------------ style pass

// ConvBlocks (non-causal; modules/commons/conv.py:84-125, prosody_util.py:299-336) on batch-indexed buffers whose
// rows [PADR, PADR+len) hold the utterance and every other row is zero.
void conan_streams::conv_blocks_noncausal(const std::string& name, int nblocks, int k, int C, Lin* x, Lin& ln, Lin& h, Lin& blkm,
                                          const TRef& npm, const int* lens, int n, int T, int& cur, hipStream_t st) {
  const float kscale = (float)std::pow((double)k, -0.5);
  for (int b = 0; b < nblocks; ++b)
    for (int j = 0; j < 2; ++j) {
      const std::string nm = name + "." + std::to_string(b) + "." + std::to_string(j);
      cnk::LNArgs l = mk_ln(x[cur].ref(PADR), ln.ref(PADR), ctx->vec(nm + ".ln.g"), ctx->vec(nm + ".ln.b"), nullptr, nullptr, n, T, C);
      l.lens = lens;
      if (j == 0) { l.mask_out = blkm.ref(PADR); l.has_mask_out = 1; }
      cnk::launch_layernorm(l, st);
      {
        ConvArgs a = mk(ctx->conv(nm + ".c1"), ln.ref(PADR), h.ref(PADR), n, T, nullptr, 1, (k - 1) / 2);
        a.slots = nullptr; a.lens = lens; a.out_scale = kscale; a.out_act = cnk::ACT_GELU;
        conv(a, st);
      }
      {
        ConvArgs a = mk(ctx->conv(nm + ".c2"), h.ref(PADR), x[cur ^ 1].ref(PADR), n, T, nullptr);
        a.slots = nullptr; a.lens = lens; a.res = x[cur].ref(PADR); a.has_res = 1; a.m1 = blkm.ref(PADR); a.has_m1 = 1;
        if (b == nblocks - 1 && j == 1) { a.m2 = npm; a.has_m2 = 1; }
        conv(a, st);
      }
      cur ^= 1;
    }
}

void conan_streams::set_reference(const int32_t* slots, int n_all, const float* ref, const int32_t* ref_len, int max_len, hipStream_t st) {
  const conan_cfg& c = ctx->cfg;
  const int H = c.hidden_size, NM = c.num_mels;
  for (int i = 0; i < n_all; ++i)
    if (ref_len[i] <= 0 || ref_len[i] > max_len || ref_len[i] > max_ref) throw Error(CONAN_ERR_INVALID, "reference length out of range");
  // (fixed-plan stream-sets: one slot per pass - the batch's longest reference sets every conv's row count, and with it the plan)
  struct StyleScope { bool& f; StyleScope(bool& x) : f(x) { f = true; } ~StyleScope() { f = false; } } style_scope(in_style_pass);
  const int batch = fixed_plan ? 1 : sp_batch;
  for (int b0 = 0; b0 < n_all; b0 += batch) {
    const int n = std::min(batch, n_all - b0);
    set_slots(slots + b0, n, st);
    std::vector<int> lens(ref_len + b0, ref_len + b0 + n), lens2(n);
    int T = 0;
    for (int i = 0; i < n; ++i) { T = std::max(T, lens[i]); lens2[i] = (lens[i] + 3) / 4; }
    const int S = (T + 3) / 4;
    pin.upload(d_lens, lens.data(), (size_t)n, st);
    pin.upload(d_lens2, lens2.data(), (size_t)n, st);
    Lin* all[] = {&s_mel, &s_np, &s_wnm, &s_x[0], &s_x[1], &s_ln, &s_h, &s_blkm, &s_wx, &s_wout, &s_win, &s_acts, &s_rs, &s_ph, &s_pm,
                  &s_px[0], &s_px[1], &s_pln, &s_phh, &s_pblk, &s_enc, &s_dots, &s_cat, &s_tok, &s_kvtmp};
    for (Lin* l : all) HIP_CHECK(hipMemsetAsync(l->base, 0, (size_t)n * l->rows * l->C * sizeof(float), st));
    {  // reference mel rows -> padded workspace
      cnk::CopyArgs ca; memset(&ca, 0, sizeof(ca));
      ca.n = n; ca.C = NM; ca.T = T; ca.lens = d_lens;
      ca.x = ch::lin_ref(const_cast<float*>(ref) + (size_t)b0 * max_len * NM, max_len, NM); ca.y = s_mel.ref(PADR);
      cnk::launch_copy_rows(ca, st);
      ca.y = s_wx.ref(PADR); cnk::launch_copy_rows(ca, st);
    }
    { cnk::RowMaskArgs a; memset(&a, 0, sizeof(a)); a.x = s_mel.ref(PADR); a.m = s_np.ref(PADR); a.lens = d_lens; a.T = T; a.n = n; a.C = NM; a.mode = 0; cnk::launch_rowmask(a, st);
      a.m = s_wnm.ref(PADR); a.mode = 1; cnk::launch_rowmask(a, st); }
    // ---- global style vector: encode_spk_embed (Conan.py:200-219)
    {
      ConvArgs a = mk(ctx->conv("conan.global_conv_in"), s_mel.ref(PADR), s_x[0].ref(PADR), n, T, nullptr);
      a.slots = nullptr; a.lens = d_lens; a.m1 = s_np.ref(PADR); a.has_m1 = 1;
      conv(a, st);
    }
    int cur = 0;
    conv_blocks_noncausal("conan.genc", 5, 31, H, s_x, s_ln, s_h, s_blkm, s_np.ref(PADR), d_lens, n, T, cur, st);
    {
      cnk::LNArgs l = mk_ln(s_x[cur].ref(PADR), s_ln.ref(PADR), ctx->vec("conan.genc.last.g"), ctx->vec("conan.genc.last.b"), nullptr, nullptr, n, T, H);
      l.lens = d_lens; l.m1 = s_np.ref(PADR); l.has_m1 = 1;
      cnk::launch_layernorm(l, st);
      ConvArgs a = mk(ctx->conv("conan.genc.post"), s_ln.ref(PADR), s_x[cur ^ 1].ref(PADR), n, T, nullptr, 1, 1);
      a.slots = nullptr; a.lens = d_lens; a.m1 = s_np.ref(PADR); a.has_m1 = 1;
      conv(a, st);
      cnk::MeanArgs m; memset(&m, 0, sizeof(m));
      m.x = s_x[cur ^ 1].ref(PADR); m.m = s_np.ref(PADR); m.out = c_style; m.out_stride = H; m.slots = d_slots; m.lens = d_lens; m.T = T; m.n = n; m.C = H;
      cnk::launch_masked_mean(m, st);
    }
    // ---- local prosody tokens: LocalStyleAdaptor (prosody_util.py:183-200)
    for (int i = 0; i < 4; ++i) {   // WN (wavenet.py:56-89): kernel 3, dilation 1, gated tanh*sigmoid
      {
        ConvArgs a = mk(ctx->conv("conan.wn.in." + std::to_string(i)), s_wx.ref(PADR), s_win.ref(PADR), n, T, nullptr, 1, 1);
        a.slots = nullptr; a.lens = d_lens; conv(a, st);
      }
      { cnk::WNGateArgs a; memset(&a, 0, sizeof(a)); a.xin = s_win.ref(PADR); a.acts = s_acts.ref(PADR); a.lens = d_lens; a.T = T; a.n = n; a.H = NM; cnk::launch_wn_gate(a, st); }
      {
        ConvArgs a = mk(ctx->conv("conan.wn.rs." + std::to_string(i)), s_acts.ref(PADR), s_rs.ref(PADR), n, T, nullptr);
        a.slots = nullptr; a.lens = d_lens; conv(a, st);
      }
      { cnk::WNUpdateArgs a; memset(&a, 0, sizeof(a)); a.rs = s_rs.ref(PADR); a.x = s_wx.ref(PADR); a.out = s_wout.ref(PADR); a.m = s_wnm.ref(PADR);
        a.lens = d_lens; a.T = T; a.n = n; a.H = NM; a.last = i == 3; a.first = i == 0; cnk::launch_wn_update(a, st); }
    }
    { cnk::PoolArgs a; memset(&a, 0, sizeof(a)); a.x = s_wout.ref(PADR); a.m = s_wnm.ref(PADR); a.y = s_px[0].ref(PADR); a.lens = d_lens; a.T = T; a.n = n; a.C = NM; a.group = 4;
      cnk::launch_group_pool(a, st); }
    { cnk::RowMaskArgs a; memset(&a, 0, sizeof(a)); a.x = s_px[0].ref(PADR); a.m = s_pm.ref(PADR); a.lens = d_lens2; a.T = S; a.n = n; a.C = NM; a.mode = 0; cnk::launch_rowmask(a, st); }
    int pc = 0;
    conv_blocks_noncausal("conan.penc", 5, 5, NM, s_px, s_pln, s_phh, s_pblk, s_pm.ref(PADR), d_lens2, n, S, pc, st);
    {
      cnk::LNArgs l = mk_ln(s_px[pc].ref(PADR), s_pln.ref(PADR), ctx->vec("conan.penc.last.g"), ctx->vec("conan.penc.last.b"), nullptr, nullptr, n, S, NM);
      l.lens = d_lens2; l.m1 = s_pm.ref(PADR); l.has_m1 = 1;
      cnk::launch_layernorm(l, st);
      ConvArgs a = mk(ctx->conv("conan.penc.post"), s_pln.ref(PADR), s_enc.ref(PADR), n, S, nullptr, 1, 1);
      a.slots = nullptr; a.lens = d_lens2; a.m1 = s_pm.ref(PADR); a.has_m1 = 1;
      conv(a, st);
    }
    // ---- VQ + positions + l1 (prosody_util.py:34-46,:88; Conan.py:244-245)
    { ConvArgs a = mk(ctx->conv("conan.vq.dot"), s_enc.ref(PADR), s_dots.ref(PADR), n, S, nullptr); a.slots = nullptr; a.lens = d_lens2; conv(a, st); }
    {
      cnk::VQArgs a; memset(&a, 0, sizeof(a));
      a.x = s_enc.ref(PADR); a.dots = s_dots.ref(PADR); a.cat = s_cat.ref(PADR); a.emb = ctx->vec("conan.vq.emb"); a.e2 = ctx->vec("conan.vq.e2");
      a.postable = ctx->vec("conan.postable"); a.ids = s_ids; a.lens = d_lens2; a.S = S; a.n = n; a.E = H; a.M = c.nvq; a.S_max = S_max;
      cnk::launch_vq(a, st);
    }
    { ConvArgs a = mk(ctx->conv("conan.l1"), s_cat.ref(PADR), s_tok.ref(PADR), n, S, nullptr); a.slots = nullptr; a.lens = d_lens2; conv(a, st); }
    { cnk::KMaskArgs a; memset(&a, 0, sizeof(a)); a.tok = s_tok.ref(PADR); a.kmask = c_kmask; a.kmask_stride = S_max; a.slots = d_slots; a.lens = d_lens2; a.S = S; a.n = n; a.S_max = S_max;
      cnk::launch_kmask(a, st); }
    cnk::launch_scatter_int(c_slen, d_slots, d_lens2, n, st);
    cnk::launch_scatter_ids(c_vqids, d_slots, s_ids, d_lens2, n, S_max, st);
    // ---- K/V of the two aligner layers, cached per slot
    for (int l = 0; l < 2; ++l) {
      TRef y; y.base = c_kv + (size_t)l * S_max * 2 * H; y.slot_stride = (long long)2 * S_max * 2 * H; y.C = 2 * H;
      y.lmask = ch::next_pow2(S_max) - 1; y.rate = 0; y.off = 0; y.mode = 0; y.pad_ = 0;
      ConvArgs a = mk(ctx->conv("conan.align." + std::to_string(l) + ".kv"), s_tok.ref(PADR), y, n, S, nullptr);
      a.slots = d_slots; a.lens = d_lens2;
      conv(a, st);
    }
    for (int i = 0; i < n; ++i) has_ref[slots[b0 + i]] = 1;
  }
}
