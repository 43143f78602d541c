// extern "C" surface of libconan_hip.so (include/conan_hip.h).
#include "streams.h"

static thread_local std::string g_err;

template <typename F>
static int guarded(F&& f) {
  try {
    f();
    // kernel launches do not return a status: a launch that the runtime rejected (resources, bad configuration) is
    // reported here instead of being lost - the product path must fail loudly
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) throw ch::Error(CONAN_ERR_HIP, std::string("HIP launch error: ") + hipGetErrorString(le));
    return CONAN_OK;
  } catch (const ch::Error& e) {
    g_err = e.what();
    return e.code;
  } catch (const std::exception& e) {
    g_err = e.what();
    return CONAN_ERR_INVALID;
  }
}

static void validate_cfg(const conan_cfg& c) {
  if (c.abi_version != CONAN_HIP_ABI_VERSION) throw Error(CONAN_ERR_INVALID, "conan_cfg.abi_version mismatch");
  if (c.models & CONAN_MODEL_HIFIGAN) {
    if (c.voc_num_ups < 1 || c.voc_num_ups > CONAN_MAX_UPS) throw Error(CONAN_ERR_INVALID, "voc_num_ups");
    if (c.voc_num_resblocks < 1 || c.voc_num_resblocks > 3) throw Error(CONAN_ERR_UNSUPPORTED, "1..3 resblock branches supported");
    if (c.voc_rb_num_dil < 1 || c.voc_rb_num_dil > CONAN_MAX_DILATIONS) throw Error(CONAN_ERR_INVALID, "voc_rb_num_dil");
    if (c.voc_upsample < 0 || c.voc_upsample > 2) throw Error(CONAN_ERR_UNSUPPORTED, "voc_upsample: 0 (shuffle), 1 (zero) or 2 (nn)");
    for (int i = 0; i < c.voc_num_ups && c.voc_upsample == 2; ++i)     // hifigan_causal.py:70-71
      if (c.voc_up_kernels[i] % 2 || c.voc_up_rates[i] < 2) throw Error(CONAN_ERR_INVALID, "upsample 'nn': kernel sizes must be even, rates >= 2");
    if (c.voc_resblock < 0 || c.voc_resblock > 2) throw Error(CONAN_ERR_UNSUPPORTED, "voc_resblock: 1 or 2");
    int ch_ = c.voc_initial_channel;
    for (int i = 0; i < c.voc_num_ups; ++i) { ch_ /= 2; if (ch_ < 4 || ch_ % 4) throw Error(CONAN_ERR_UNSUPPORTED, "vocoder channel ladder must stay a multiple of 4"); }
    if (c.num_mels % 4) throw Error(CONAN_ERR_UNSUPPORTED, "num_mels must be a multiple of 4");
  }
  if (c.models & CONAN_MODEL_EMFORMER) {
    if (c.emf_input_dim % c.emf_heads || c.emf_input_dim / c.emf_heads > 16) throw Error(CONAN_ERR_UNSUPPORTED, "emformer head_dim must be <= 16");
    if (c.emf_input_dim % 4 || c.emf_input_dim > 512) throw Error(CONAN_ERR_UNSUPPORTED, "emformer input_dim");
    if (c.emf_segment < 1 || c.emf_right_context < 0) throw Error(CONAN_ERR_INVALID, "emformer segment/right context");
    if (c.emf_max_memory_size < 0) throw Error(CONAN_ERR_INVALID, "emf_max_memory_size");
    if (c.emf_max_memory_size > 32) throw Error(CONAN_ERR_UNSUPPORTED, "emf_max_memory_size > 32");
    if (c.emf_max_memory_size + c.emf_right_context + c.emf_left_context + c.emf_segment > 128 || c.emf_heads > 16) throw Error(CONAN_ERR_UNSUPPORTED, "emformer attention supports <= 128 keys, <= 16 heads");
  }
  if (c.models & CONAN_MODEL_CONAN) {
    if (c.hidden_size % 8 || c.hidden_size > 512) throw Error(CONAN_ERR_UNSUPPORTED, "hidden_size must be a multiple of 8 and <= 512");
    if (c.dec_num_blocks < 1 || c.dec_num_blocks > CONAN_MAX_DEC_BLOCKS) throw Error(CONAN_ERR_INVALID, "dec_num_blocks");
    if (c.nvq < 1) throw Error(CONAN_ERR_INVALID, "nvq");
  }
}

extern "C" {

const char* conan_last_error(void) { return g_err.c_str(); }
int conan_abi_version(void) { return CONAN_HIP_ABI_VERSION; }

int conan_ctx_create(int device, const conan_cfg* cfg, conan_ctx** out) {
  return guarded([&] {
    if (!cfg || !out) throw Error(CONAN_ERR_INVALID, "null argument");
    validate_cfg(*cfg);
    int ndev = 0;
    HIP_CHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) throw Error(CONAN_ERR_HIP, "no such HIP device (libconan_hip has no CPU fallback)");
    HIP_CHECK(hipSetDevice(device));
    conan_ctx* c = new conan_ctx();
    c->device = device;
    c->cfg = *cfg;
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    int hop = 1;
    for (int i = 0; i < cfg->voc_num_ups; ++i) hop *= cfg->voc_up_rates[i];
    c->hop = hop;
    *out = c;
  });
}

int conan_ctx_destroy(conan_ctx* ctx) {
  return guarded([&] { delete ctx; });
}

int conan_ctx_load_tensor(conan_ctx* ctx, const char* key, const float* host, const int64_t* shape, int ndim) {
  int known = 1;
  int rc = guarded([&] {
    if (!ctx || !key || !host || ndim < 0 || ndim > 4) throw Error(CONAN_ERR_INVALID, "bad argument");
    if (ctx->finalized) throw Error(CONAN_ERR_STATE, "context already finalized");
    std::string k(key);
    if (k.rfind("emformer.", 0) != 0 && k.rfind("conan.", 0) != 0 && k.rfind("hifigan.", 0) != 0) { known = 0; return; }
    ch::HostTensor t;
    t.shape.assign(shape, shape + ndim);
    int64_t n = t.numel();
    if (n < 0 || n > (int64_t)1 << 31) throw Error(CONAN_ERR_SHAPE, "tensor too large");
    t.data.assign(host, host + n);
    ctx->raw[k] = std::move(t);
  });
  if (rc != CONAN_OK) return rc;
  return known ? 0 : 1;
}

int conan_ctx_finalize(conan_ctx* ctx) {
  return guarded([&] {
    if (!ctx) throw Error(CONAN_ERR_INVALID, "null ctx");
    if (ctx->finalized) return;
    HIP_CHECK(hipSetDevice(ctx->device));
    if (ctx->cfg.models & CONAN_MODEL_HIFIGAN) ctx->finalize_hifigan();
    if (ctx->cfg.models & CONAN_MODEL_EMFORMER) ctx->finalize_emformer();
    if (ctx->cfg.models & CONAN_MODEL_CONAN) ctx->finalize_conan();
    ctx->raw.clear();
    ctx->finalized = true;
  });
}

int conan_streams_create(conan_ctx* ctx, int max_slots, int max_frames, int max_ref_frames, conan_streams** out) {
  return conan_streams_create_opts(ctx, max_slots, max_frames, max_ref_frames, nullptr, out);
}

int conan_streams_arith(const conan_streams* s) {
  if (!s) { g_err = "null streams"; return CONAN_ERR_INVALID; }
  return s->rb_limb ? CONAN_ARITH_LIMB : CONAN_ARITH_F32;
}

int conan_streams_create_opts(conan_ctx* ctx, int max_slots, int max_frames, int max_ref_frames, const conan_streams_opts* opts,
                              conan_streams** out) {
  return guarded([&] {
    if (!ctx || !out) throw Error(CONAN_ERR_INVALID, "null argument");
    int arith = CONAN_ARITH_AUTO, flags = 0;
    if (opts) {
      if (opts->abi_version != CONAN_HIP_ABI_VERSION) throw Error(CONAN_ERR_INVALID, "conan_streams_opts.abi_version mismatch");
      for (int r : opts->reserved) if (r != 0) throw Error(CONAN_ERR_INVALID, "conan_streams_opts.reserved must be 0");
      arith = opts->arith; flags = opts->flags;
      if (opts->reserved0 != 0) throw Error(CONAN_ERR_INVALID, "conan_streams_opts.reserved0 must be 0");
      if (flags & ~(CONAN_STREAMS_FUSED_DECODER_BLOCKS | CONAN_STREAMS_SEPARATE_SMALL_STEPS | CONAN_STREAMS_FIXED_PLAN | CONAN_STREAMS_SHARED_DEVICE)) throw Error(CONAN_ERR_INVALID, "conan_streams_opts.flags: unknown bits");
      if (arith != CONAN_ARITH_AUTO && arith != CONAN_ARITH_F32 && arith != CONAN_ARITH_LIMB) throw Error(CONAN_ERR_INVALID, "conan_streams_opts.arith: 0 (auto), 1 (f32) or 2 (limb)");
    }
    if (!ctx->finalized) throw Error(CONAN_ERR_STATE, "conan_ctx_finalize must run before conan_streams_create");
    if (max_slots < 1 || max_frames < 1) throw Error(CONAN_ERR_INVALID, "max_slots / max_frames");
    HIP_CHECK(hipSetDevice(ctx->device));
    conan_streams* s = new conan_streams();
    try {
      s->parse_dev_plan(opts ? opts->dev_plan : nullptr);
      s->fixed_plan = (flags & CONAN_STREAMS_FIXED_PLAN) != 0; s->shared_device = (flags & CONAN_STREAMS_SHARED_DEVICE) != 0;
      s->ctx = ctx; s->live = &device_live_streams(ctx->device); s->live->fetch_add(1); s->max_slots = max_slots; s->max_frames = std::max(max_frames, ctx->cfg.emf_segment); s->max_ref = std::max(4, max_ref_frames);
      s->d_slots = (int*)s->alloc(max_slots + cnk::kSlotTablePad); s->d_ident = (int*)s->alloc(max_slots + 1); s->d_zero = (int*)s->alloc(max_slots);
      s->d_lens = (int*)s->alloc(max_slots); s->d_lens2 = (int*)s->alloc(max_slots);
      s->d_codes = (int*)s->alloc((size_t)max_slots * s->max_frames * 2);
      s->sk_slab_floats = 8ll << 20; s->sk_max_tiles = 4096;
      for (int w = 0; w < 3; ++w) { s->sk_slab[w] = s->alloc((size_t)s->sk_slab_floats); s->sk_counters[w] = (int*)s->alloc(s->sk_max_tiles); }
      for (int w = 0; w < 2; ++w) s->rb_sched[w] = (int*)s->alloc(4);
      for (int w = 0; w < 3; ++w) s->cp_ticket[w] = (int*)s->alloc(4);
      { const char* e = s->dev("RESERVE_CUS"); s->reserve_cus = e ? atoi(e) : 0; }
      { const char* e = s->dev("ROWCONV"); s->use_rowconv = !(e && e[0] == '0'); }
      s->rb_merge = s->dev("RB_NOMERGE") == nullptr;
      // fp32 products of the vocoder's matrix kernels as six bf16 limb products (resblock_limb.hip, conv_limb.hip) or on the
      // f32-input MFMA: conan_streams_opts.arith.  AUTO = the limb form wherever the context packed limb weights (ResBlock1
      // vocoders); the developer switch CONAN_RB_NOLIMB=1 turns AUTO into F32 for A/B runs - it never overrides an explicit request.
      if (arith == CONAN_ARITH_LIMB && !((ctx->cfg.models & CONAN_MODEL_HIFIGAN) && ctx->has_limb_weights))
        throw Error(CONAN_ERR_UNSUPPORTED, "arith = limb: this context holds no bf16-limb weights (no HiFi-GAN model, or a vocoder configuration without limb kernels)");
      s->arith_auto = arith == CONAN_ARITH_AUTO;
      s->rb_limb = arith == CONAN_ARITH_LIMB || (arith == CONAN_ARITH_AUTO && ctx->has_limb_weights && s->dev("RB_NOLIMB") == nullptr);
      // deployment flags (conan_streams_opts.flags)
      s->opt_flags = flags;
      s->mega_single = !(s->opt_flags & CONAN_STREAMS_SEPARATE_SMALL_STEPS);
      { const char* e = s->dev("FENCED"); s->fenced = e && e[0] == '1'; }
      { const char* e = s->dev("DEC_MEGA"); s->use_mega = !(e && e[0] == '0'); }
      { const char* e = s->dev("MEGA_GRID"); if (e && atoi(e) > 0) s->mega_grid = std::min(atoi(e), ctx->num_cu); }
      // (a CU-masked front-end stream - developer switch - cannot hold the megakernel's grid resident: its barriers would never complete)
      { const char* e = s->dev("FRONT_CUSTRIDE"); if (e && atoi(e) >= 2) s->use_mega = false; }
      { const char* e = s->dev("MEGA_GS"); if (e && (atoi(e) == 4 || atoi(e) == 8 || atoi(e) == 16)) s->mega_gs = atoi(e); }
      { const char* e = s->dev("MEGA_NARROW"); if (e && e[0] == '0') s->mega_narrow_ksplit = false; }      // developer A/B switch
      s->mega_bar = reinterpret_cast<unsigned*>(s->alloc(16 * (size_t)(ctx->num_cu + 2)));
      s->mega_x = reinterpret_cast<unsigned*>(s->alloc(256 + 32 * 64));

      {  // guard block of the bounded waits: [0] code, [2..3] device address of the host-mapped copy
        HIP_CHECK(hipHostMalloc((void**)&s->h_guard, 64, hipHostMallocMapped));
        memset(s->h_guard, 0, 64);
        unsigned* hdev = nullptr;
        HIP_CHECK(hipHostGetDevicePointer((void**)&hdev, s->h_guard, 0));
        s->d_guard = reinterpret_cast<unsigned*>(s->alloc(16));
        HIP_CHECK(hipMemcpy(s->d_guard + 2, &hdev, sizeof(hdev), hipMemcpyHostToDevice));
      }
      s->slot_seen.assign(max_slots, 0); s->has_ref.assign(max_slots, 0); s->voc_fresh.assign(max_slots, 1);
      s->pin.init((size_t)max_slots + cnk::kSlotTablePad);
      s->pos_emf = (int*)s->alloc(max_slots); s->pos_dec = (int*)s->alloc(max_slots); s->pos_voc = (int*)s->alloc(max_slots);
      std::vector<int> id(max_slots);
      for (int i = 0; i < max_slots; ++i) id[i] = i;
      HIP_CHECK(hipMemcpy(s->d_ident, id.data(), max_slots * sizeof(int), hipMemcpyHostToDevice));
      if (ctx->cfg.models & CONAN_MODEL_HIFIGAN) s->build_vocoder();
      if (ctx->cfg.models & CONAN_MODEL_EMFORMER) s->build_emformer();
      if (ctx->cfg.models & CONAN_MODEL_CONAN) s->build_decoder();
    } catch (...) { delete s; throw; }
    *out = s;
  });
}

int conan_streams_destroy(conan_streams* s) {
  return guarded([&] { if (s) { (void)hipDeviceSynchronize(); delete s; } });
}

int conan_streams_reset(conan_streams* s, const int32_t* slots, int n, int which, void* stream) {
  return guarded([&] {
    if (!s || !slots) throw Error(CONAN_ERR_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK(hipSetDevice(s->ctx->device)); s->check_fault();
    s->join((hipStream_t)stream);
    s->set_slots(slots, n, st);
    const int models = s->ctx->cfg.models & which;
    auto zero = [&](std::vector<std::pair<float*, long long>>& v, int* pos) {
      for (auto& b : v) cnk::launch_zero_slots(b.first, b.second, b.second, s->d_slots, n, st);
      cnk::launch_fill_int(pos, s->d_slots, n, 0, st);
    };
    if (models & CONAN_MODEL_HIFIGAN) {
      zero(s->voc_state, s->pos_voc);
      for (int i = 0; i < n; ++i) s->voc_fresh[slots[i]] = 1;
    }
    if (models & CONAN_MODEL_EMFORMER) zero(s->emf_state, s->pos_emf);
    if (models & CONAN_MODEL_CONAN) zero(s->dec_state, s->pos_dec);
  });
}

int conan_set_reference(conan_streams* s, const int32_t* slots, int n, const float* ref_mel_dev, const int32_t* ref_len,
                        int max_len, void* stream) {
  return guarded([&] {
    if (!s || !slots || !ref_mel_dev || !ref_len) throw Error(CONAN_ERR_INVALID, "null argument (the reference raises ValueError when ref is None)");
    if (!(s->ctx->cfg.models & CONAN_MODEL_CONAN)) throw Error(CONAN_ERR_STATE, "context holds no Conan model");
    if (n < 1 || n > s->max_slots) throw Error(CONAN_ERR_INVALID, "slot count out of range");
    HIP_CHECK(hipSetDevice(s->ctx->device)); s->check_fault();
    s->join((hipStream_t)stream);
    s->set_reference(slots, n, ref_mel_dev, ref_len, max_len, (hipStream_t)stream);
  });
}

int conan_emformer_step(conan_streams* s, const int32_t* slots, int n, const float* chunk_dev, float* out_dev, float* logits_dev,
                        int32_t* codes_dev, void* stream) {
  return guarded([&] {
    if (!s || !slots || !chunk_dev) throw Error(CONAN_ERR_INVALID, "null argument");
    if (!(s->ctx->cfg.models & CONAN_MODEL_EMFORMER)) throw Error(CONAN_ERR_STATE, "context holds no Emformer model");
    HIP_CHECK(hipSetDevice(s->ctx->device)); s->check_fault();
    s->join((hipStream_t)stream);
    s->set_slots(slots, n, (hipStream_t)stream);
    s->emformer_step(n, chunk_dev, out_dev, logits_dev, codes_dev, (hipStream_t)stream);
  });
}

int conan_emformer_head_dim(conan_streams* s, const char* head) {
  int k = 0;
  const int rc = guarded([&] {
    if (!s || !head) throw Error(CONAN_ERR_INVALID, "null argument");
    auto it = s->ctx->convs.find(std::string("emf.head.") + head);
    k = it == s->ctx->convs.end() ? 0 : it->second.Cout;
  });
  return rc < 0 ? rc : k;
}

int conan_emformer_project(conan_streams* s, const char* head, const float* x_dev, int rows, float* y_dev, void* stream) {
  return guarded([&] {
    if (!s || !head || !x_dev || !y_dev) throw Error(CONAN_ERR_INVALID, "null argument");
    if (!(s->ctx->cfg.models & CONAN_MODEL_EMFORMER)) throw Error(CONAN_ERR_STATE, "context holds no Emformer model");
    if (rows <= 0) return;
    const std::string name = std::string("emf.head.") + head;
    if (!s->ctx->convs.count(name)) throw Error(CONAN_ERR_MISSING, std::string("the Emformer checkpoint holds no output head '") + head + "'");
    HIP_CHECK(hipSetDevice(s->ctx->device)); s->check_fault();
    hipStream_t st = (hipStream_t)stream;
    s->join(st);
    const ch::PackedConv& pc = s->ctx->conv(name);
    // a Linear is a k = 1 conv over one "slot" of `rows` rows (conv_mfma; no ring, no slot table)
    s->conv(s->mk(pc, ch::lin_ref(const_cast<float*>(x_dev), rows, pc.Cin), ch::lin_ref(y_dev, rows, pc.Cout), 1, rows, nullptr), st);
  });
}

int conan_decoder_step(conan_streams* s, const int32_t* slots, int n, int frames, const int32_t* codes_dev, float* mel_out_dev,
                       float* uv_pred_dev, float* f0_dev, int32_t* bins_dev, float* decoder_inp_dev, void* stream) {
  return guarded([&] {
    if (!s || !slots || !codes_dev || !mel_out_dev) throw Error(CONAN_ERR_INVALID, "null argument");
    if (!(s->ctx->cfg.models & CONAN_MODEL_CONAN)) throw Error(CONAN_ERR_STATE, "context holds no Conan model");
    if (frames < 1 || frames > s->max_frames) throw Error(CONAN_ERR_INVALID, "frames out of range");
    HIP_CHECK(hipSetDevice(s->ctx->device)); s->check_fault();
    s->join((hipStream_t)stream);
    s->set_slots(slots, n, (hipStream_t)stream);
    conan_decoder_taps taps; memset(&taps, 0, sizeof(taps));
    taps.uv_pred = uv_pred_dev; taps.f0_denorm_pred = f0_dev; taps.pitch_bins = bins_dev; taps.decoder_inp = decoder_inp_dev;
    s->decoder_step(n, frames, codes_dev, mel_out_dev, taps, (hipStream_t)stream);
  });
}

int conan_decoder_step_taps(conan_streams* s, const int32_t* slots, int n, int frames, const int32_t* codes_dev, float* mel_out_dev,
                            const conan_decoder_taps* taps, void* stream) {
  return guarded([&] {
    if (!s || !slots || !codes_dev || !mel_out_dev) throw Error(CONAN_ERR_INVALID, "null argument");
    if (!(s->ctx->cfg.models & CONAN_MODEL_CONAN)) throw Error(CONAN_ERR_STATE, "context holds no Conan model");
    if (frames < 1 || frames > s->max_frames) throw Error(CONAN_ERR_INVALID, "frames out of range");
    HIP_CHECK(hipSetDevice(s->ctx->device)); s->check_fault();
    s->join((hipStream_t)stream);
    s->set_slots(slots, n, (hipStream_t)stream);
    conan_decoder_taps none; memset(&none, 0, sizeof(none));
    s->decoder_step(n, frames, codes_dev, mel_out_dev, taps ? *taps : none, (hipStream_t)stream);
  });
}

int conan_get_style(conan_streams* s, const int32_t* slots, int n, float* style_dev, int32_t* max_tokens_out, void* stream) {
  return guarded([&] {
    if (!s || !slots || !style_dev) throw Error(CONAN_ERR_INVALID, "null argument");
    if (!(s->ctx->cfg.models & CONAN_MODEL_CONAN)) throw Error(CONAN_ERR_STATE, "context holds no Conan model");
    if (n < 1 || n > s->max_slots) throw Error(CONAN_ERR_INVALID, "slot count out of range");
    for (int i = 0; i < n; ++i) {
      if (slots[i] < 0 || slots[i] >= s->max_slots) throw Error(CONAN_ERR_INVALID, "slot index out of range");
      if (!s->has_ref[slots[i]]) throw Error(CONAN_ERR_STATE, "conan_get_style before conan_set_reference");
    }
    HIP_CHECK(hipSetDevice(s->ctx->device)); s->check_fault();
    s->join((hipStream_t)stream);
    const int H = s->ctx->cfg.hidden_size;
    for (int i = 0; i < n; ++i) {
      if (slots[i] < 0 || slots[i] >= s->max_slots) throw Error(CONAN_ERR_INVALID, "slot index out of range");
      HIP_CHECK(hipMemcpyAsync(style_dev + (size_t)i * H, s->c_style + (size_t)slots[i] * H, (size_t)H * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    if (max_tokens_out) *max_tokens_out = s->S_max;
  });
}

int conan_set_style(conan_streams* s, const int32_t* slots, int n, const float* style_dev, void* stream) {
  return guarded([&] {
    if (!s || !slots || !style_dev) throw Error(CONAN_ERR_INVALID, "null argument");
    if (!(s->ctx->cfg.models & CONAN_MODEL_CONAN)) throw Error(CONAN_ERR_STATE, "context holds no Conan model");
    HIP_CHECK(hipSetDevice(s->ctx->device)); s->check_fault();
    s->join((hipStream_t)stream);
    s->set_slots(slots, n, (hipStream_t)stream);
    for (int i = 0; i < n; ++i)
      if (!s->has_ref[slots[i]]) throw Error(CONAN_ERR_STATE, "conan_set_style before conan_set_reference (the prosody tokens come from the reference mel)");
    cnk::launch_scatter_rows(s->c_style, style_dev, s->d_slots, n, s->ctx->cfg.hidden_size, (hipStream_t)stream);
  });
}

int conan_get_prosody_ids(conan_streams* s, const int32_t* slots, int n, int32_t* ids_dev, int32_t* count_dev, void* stream) {
  return guarded([&] {
    if (!s || !slots || !ids_dev) throw Error(CONAN_ERR_INVALID, "null argument");
    if (!(s->ctx->cfg.models & CONAN_MODEL_CONAN)) throw Error(CONAN_ERR_STATE, "context holds no Conan model");
    HIP_CHECK(hipSetDevice(s->ctx->device)); s->check_fault();
    s->join((hipStream_t)stream);
    s->set_slots(slots, n, (hipStream_t)stream);
    for (int i = 0; i < n; ++i)
      if (!s->has_ref[slots[i]]) throw Error(CONAN_ERR_STATE, "conan_get_prosody_ids before conan_set_reference");
    cnk::launch_gather_ids(ids_dev, count_dev, s->c_vqids, s->c_slen, s->d_slots, n, s->S_max, (hipStream_t)stream);
  });
}

int conan_hifigan_step_taps(conan_streams* s, const int32_t* slots, int n, int frames, const float* mel_dev, float* wav_out_dev,
                            float* pre_tanh_dev, const conan_hifigan_taps* taps, void* stream) {
  return guarded([&] {
    if (!s || !slots || !mel_dev || !wav_out_dev) throw Error(CONAN_ERR_INVALID, "null argument");
    if (!(s->ctx->cfg.models & CONAN_MODEL_HIFIGAN)) throw Error(CONAN_ERR_STATE, "context holds no HiFi-GAN model");
    if (frames < 1 || frames > s->max_frames) throw Error(CONAN_ERR_INVALID, "frames out of range");
    HIP_CHECK(hipSetDevice(s->ctx->device)); s->check_fault();
    s->join((hipStream_t)stream);
    s->set_slots(slots, n, (hipStream_t)stream);
    s->hifigan_step(n, frames, mel_dev, wav_out_dev, pre_tanh_dev, (hipStream_t)stream, taps);
  });
}

int conan_hifigan_step(conan_streams* s, const int32_t* slots, int n, int frames, const float* mel_dev, float* wav_out_dev,
                       float* pre_tanh_dev, void* stream) {
  return guarded([&] {
    if (!s || !slots || !mel_dev || !wav_out_dev) throw Error(CONAN_ERR_INVALID, "null argument");
    if (!(s->ctx->cfg.models & CONAN_MODEL_HIFIGAN)) throw Error(CONAN_ERR_STATE, "context holds no HiFi-GAN model");
    if (frames < 1 || frames > s->max_frames) throw Error(CONAN_ERR_INVALID, "frames out of range");
    HIP_CHECK(hipSetDevice(s->ctx->device)); s->check_fault();
    s->join((hipStream_t)stream);
    s->set_slots(slots, n, (hipStream_t)stream);
    s->hifigan_step(n, frames, mel_dev, wav_out_dev, pre_tanh_dev, (hipStream_t)stream);
  });
}

int conan_step(conan_streams* s, const int32_t* slots, int n, int emit, const float* mel_chunk_dev, int32_t* codes_dev,
               float* mel_out_dev, float* wav_out_dev, void* stream) {
  return guarded([&] {
    if (!s || !slots || !mel_chunk_dev || !wav_out_dev) throw Error(CONAN_ERR_INVALID, "null argument");
    const int all = CONAN_MODEL_EMFORMER | CONAN_MODEL_CONAN | CONAN_MODEL_HIFIGAN;
    if ((s->ctx->cfg.models & all) != all) throw Error(CONAN_ERR_STATE, "conan_step needs all three models in the context");
    if (s->ctx->cfg.voc_upsample == 2) throw Error(CONAN_ERR_UNSUPPORTED, "fused chunk steps carry vocoder state from chunk to chunk; upsample 'nn' (CausalUpsampleBlock1) looks ahead: "
                                                                        "step the Emformer and decoder per chunk and run conan_hifigan_step over the mel prefix after a reset (inference/Conan.py:147-155)");
    const int seg = s->ctx->cfg.emf_segment;
    if (emit < 1 || emit > seg) throw Error(CONAN_ERR_INVALID, "emit must be in [1, segment]");
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK(hipSetDevice(s->ctx->device)); s->check_fault();
    s->join((hipStream_t)stream);
    s->set_slots(slots, n, st);
    int* codes_seg = codes_dev ? codes_dev : s->d_codes;
    s->emformer_step(n, mel_chunk_dev, nullptr, nullptr, codes_seg, st);
    const int* codes_emit = codes_seg;
    if (emit != seg && n > 1) {
      int* compact = s->d_codes + (size_t)s->max_slots * s->max_frames;
      cnk::launch_copy_int_rows(compact, codes_seg, n, emit, seg, st);
      codes_emit = compact;
    }
    float* mel = mel_out_dev ? mel_out_dev : s->c_mel.base;
    { conan_decoder_taps none; memset(&none, 0, sizeof(none)); s->decoder_step(n, emit, codes_emit, mel, none, st); }
    s->hifigan_step(n, emit, mel, wav_out_dev, nullptr, st);
  });
}

// Pipelined variant of conan_step.  Within one stream-set the three stages of a chunk are strictly ordered, but the
// front-end of the next chunk depends only on front-end state, so it runs on its own HIP stream while the vocoder of
// this chunk is still busy on another: the ~60 latency-bound front-end launches fill the gaps of the vocoder's large
// kernels instead of adding to the step time.
int conan_step_async(conan_streams* s, const int32_t* slots, int n, int emit, const float* mel_chunk_dev, int32_t* codes_dev,
                     float* mel_out_dev, float* wav_out_dev, void* stream) {
  return guarded([&] {
    if (!s || !slots || !mel_chunk_dev || !wav_out_dev) throw Error(CONAN_ERR_INVALID, "null argument");
    const int all = CONAN_MODEL_EMFORMER | CONAN_MODEL_CONAN | CONAN_MODEL_HIFIGAN;
    if ((s->ctx->cfg.models & all) != all) throw Error(CONAN_ERR_STATE, "conan_step_async needs all three models in the context");
    if (s->ctx->cfg.voc_upsample == 2) throw Error(CONAN_ERR_UNSUPPORTED, "fused chunk steps carry vocoder state from chunk to chunk; upsample 'nn' (CausalUpsampleBlock1) looks ahead: "
                                                                        "step the Emformer and decoder per chunk and run conan_hifigan_step over the mel prefix after a reset (inference/Conan.py:147-155)");
    const int seg = s->ctx->cfg.emf_segment;
    if (emit < 1 || emit > seg) throw Error(CONAN_ERR_INVALID, "emit must be in [1, segment]");
    if (s->prof_on) throw Error(CONAN_ERR_STATE, "profiling is not available for pipelined steps");
    HIP_CHECK(hipSetDevice(s->ctx->device)); s->check_fault();
    s->async_init();
    const long long t = s->async_steps;
    constexpr int NP = conan_streams::NP;
    const int p = (int)(t % NP), pl = (int)((t + NP - 1) % NP);      // hand-off ring positions of this step and of the previous one
    // Three stages on three internal streams: Emformer(t) -> codes, decoder(t) -> mel, vocoder(t) -> audio.  With steps
    // issued back to back the stages work on consecutive chunks at the same time (Emformer of chunk t+2 beside the decoder
    // of t+1 beside the vocoder of t): the decoder's ~50 latency-bound launches no longer queue behind the Emformer's
    // one long launch, and their tail no longer leaves the vocoder stream idle.
    // An EMPTY pipeline (the first step, or every earlier step has completed - e.g. behind the caller's join + synchronize): nothing of
    // this stream-set can overlap this step's Emformer launch, which the decoder and vocoder of the same chunk wait for - it may take
    // the blocking steps' launch shape (one workgroup per CU where the stream-set is alone on the device: 136 instead of 190 us on
    // the first chunk's critical path; the feed-forward's sum does not depend on the cluster size, so the bits are the same).
    s->pipe_idle = t == 0 || (hipEventQuery(s->ev_emf[pl]) == hipSuccess && hipEventQuery(s->ev_front[pl]) == hipSuccess && hipEventQuery(s->ev_voc[pl]) == hipSuccess);
    (void)hipGetLastError();      // (hipErrorNotReady from a query is an answer, not an error)
    // inputs are ready in the caller's stream order
    HIP_CHECK(hipEventRecord(s->ev_in[p], (hipStream_t)stream));
    HIP_CHECK(hipStreamWaitEvent(s->st_emf, s->ev_in[p], 0));
    const bool tl = s->tl_on && s->tl_n < (int)s->tl_ev.size() / 6;
    hipEvent_t* te = tl ? &s->tl_ev[(size_t)s->tl_n * 6] : nullptr;
    // a changed slot list rewrites the table the in-flight decoder / vocoder still read: drain them first
    bool same = (int)s->h_slots.size() == n;
    for (int i = 0; same && i < n; ++i) same = s->h_slots[i] == slots[i];
    if (!same && t >= 1) {
      HIP_CHECK(hipStreamWaitEvent(s->st_emf, s->ev_front[pl], 0));
      HIP_CHECK(hipStreamWaitEvent(s->st_emf, s->ev_voc[pl], 0));
    }
    s->set_slots(slots, n, s->st_emf);
    // the code buffer at this ring position is free once the decoder of step t-NP has read it: the Emformer may run
    // NP steps ahead of the decoder (it is dispatched late - its 129 KB of LDS per block only fit on CUs that a vocoder
    // launch has left - so the decoder must not have to wait for the Emformer of its own chunk)
    if (t >= NP) HIP_CHECK(hipStreamWaitEvent(s->st_emf, s->ev_front[p], 0));
    int* codes_seg = s->codes_hand[p];
    // developer timing switch (results are then meaningless): CONAN_SKIP_STAGE bit 0 skips the Emformer launch, bit 1 the decoder's
#ifdef CONAN_DEV_SWITCHES        // `make DEV=1`: timing experiments only, never in the shipped library (a skipped stage returns garbage with CONAN_OK)
    static const int skip = ch::dev_getenv("CONAN_SKIP_STAGE") ? atoi(ch::dev_getenv("CONAN_SKIP_STAGE")) : 0;
    // (the Emformer's workgroups need whole CUs for ~0.15 ms; they are kept away from the pair kernel's launches: see ev_wide)
    static const bool hold = ch::dev_getenv("CONAN_EMF_HOLD") != nullptr;      // (off by default: see streams.h, ev_wide)
#else
    constexpr int skip = 0; constexpr bool hold = false;
#endif
    if (hold && t >= 2 && s->ev_wide[(t + NP - 2) % NP] && s->wide_marked[(t + NP - 2) % NP]) HIP_CHECK(hipStreamWaitEvent(s->st_emf, s->ev_wide[(t + NP - 2) % NP], 0));
    if (tl) HIP_CHECK(hipEventRecord(te[0], s->st_emf));
    if (!(skip & 1)) s->emformer_step(n, mel_chunk_dev, nullptr, nullptr, codes_seg, s->st_emf);
    if (tl) HIP_CHECK(hipEventRecord(te[1], s->st_emf));
    HIP_CHECK(hipEventRecord(s->ev_emf[p], s->st_emf));
    HIP_CHECK(hipStreamWaitEvent(s->st_front, s->ev_emf[p], 0));
    // the mel hand-off buffer at this ring position is free once the vocoder of step t-NP has copied it into its ring
    if (t >= NP) HIP_CHECK(hipStreamWaitEvent(s->st_front, s->ev_voc[p], 0));
    if (tl) HIP_CHECK(hipEventRecord(te[2], s->st_front));
    // the caller's copies of the step's codes and mel frames travel with the decoder step (operators of its one launch)
    conan_streams::DecExtra ex;
    if (codes_dev) { ex.codes_dst = codes_dev; ex.codes_src = codes_seg; ex.codes_words = n * seg; }
    ex.mel_out2 = mel_out_dev;
    const int* codes_emit = codes_seg;
    if (emit != seg && n > 1) {
      int* compact = s->d_codes + (size_t)s->max_slots * s->max_frames;
      cnk::launch_copy_int_rows(compact, codes_seg, n, emit, seg, s->st_front);
      codes_emit = compact;
    }
    float* mel = s->mel_hand[p];
    if (!(skip & 2)) { conan_decoder_taps none; memset(&none, 0, sizeof(none)); s->decoder_step(n, emit, codes_emit, mel, none, s->st_front, &ex); }
    if (tl) HIP_CHECK(hipEventRecord(te[3], s->st_front));
    HIP_CHECK(hipEventRecord(s->ev_front[p], s->st_front));
    HIP_CHECK(hipStreamWaitEvent(s->st_voc, s->ev_front[p], 0));
    if (s->fence_set) {      // the caller's output fence: only the stage that writes the audio buffer waits for it
      if (s->fence_event) HIP_CHECK(hipStreamWaitEvent(s->st_voc, s->fence_event, 0));
      else {
        HIP_CHECK(hipEventRecord(s->ev_fence[p], s->fence_stream));
        HIP_CHECK(hipStreamWaitEvent(s->st_voc, s->ev_fence[p], 0));
      }
      s->fence_set = false; s->fence_event = nullptr;
    }
    if (tl) HIP_CHECK(hipEventRecord(te[4], s->st_voc));
    if (!s->ev_wide[p]) HIP_CHECK(hipEventCreateWithFlags(&s->ev_wide[p], hipEventDisableTiming));
    s->mark_wide = s->ev_wide[p];
    s->wide_marked[p] = false;
    s->hifigan_step(n, emit, mel, wav_out_dev, nullptr, s->st_voc);
    s->mark_wide = nullptr;
    if (tl) { HIP_CHECK(hipEventRecord(te[5], s->st_voc)); s->tl_n++; }
    HIP_CHECK(hipEventRecord(s->ev_voc[p], s->st_voc));
    if (s->clock_on && s->clock_n < (int)s->clock_ev.size()) HIP_CHECK(hipEventRecord(s->clock_ev[s->clock_n++], s->st_voc));   // step completion stamp
    s->async_steps = t + 1;
  });
}

int conan_step_clock(conan_streams* s, int capacity) {
  return guarded([&] {
    if (!s || capacity < 0 || capacity > 4096) throw Error(CONAN_ERR_INVALID, "conan_step_clock: capacity in [0, 4096]");
    HIP_CHECK(hipSetDevice(s->ctx->device)); s->check_fault();
    while ((int)s->clock_ev.size() < capacity) { hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); s->clock_ev.push_back(e); }
    s->clock_on = capacity > 0; s->clock_n = 0;
  });
}

int conan_step_timeline(conan_streams* s, int capacity) {
  return guarded([&] {
    if (!s || capacity < 0 || capacity > 1024) throw Error(CONAN_ERR_INVALID, "conan_step_timeline: capacity in [0, 1024]");
    HIP_CHECK(hipSetDevice(s->ctx->device)); s->check_fault();
    while ((int)s->tl_ev.size() < capacity * 6) { hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); s->tl_ev.push_back(e); }
    s->tl_on = capacity > 0; s->tl_n = 0;
  });
}

int conan_step_timeline_read(conan_streams* s, double* ms_out, int cap_steps) {
  int cnt = 0;
  const int rc = guarded([&] {
    if (!s || (!ms_out && cap_steps > 0)) throw Error(CONAN_ERR_INVALID, "null argument");
    if (s->tl_n < 1) return;
    HIP_CHECK(hipEventSynchronize(s->tl_ev[(size_t)s->tl_n * 6 - 1]));
    HIP_CHECK(hipDeviceSynchronize());
    for (int i = 0; i < s->tl_n && cnt < cap_steps; ++i, ++cnt)
      for (int e = 0; e < 6; ++e) {
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, s->tl_ev[0], s->tl_ev[(size_t)i * 6 + e]));
        ms_out[(size_t)cnt * 6 + e] = ms;
      }
  });
  return rc < 0 ? rc : cnt;
}

int conan_step_clock_read(conan_streams* s, double* ms_out, int cap) {
  int cnt = 0;
  const int rc = guarded([&] {
    if (!s || (!ms_out && cap > 0)) throw Error(CONAN_ERR_INVALID, "null argument");
    if (s->clock_n < 2) return;
    HIP_CHECK(hipEventSynchronize(s->clock_ev[s->clock_n - 1]));
    for (int i = 0; i + 1 < s->clock_n && cnt < cap; ++i) {
      float ms = 0.f;
      HIP_CHECK(hipEventElapsedTime(&ms, s->clock_ev[i], s->clock_ev[i + 1]));
      ms_out[cnt++] = ms;
    }
  });
  return rc < 0 ? rc : cnt;
}

int conan_streams_test_fault(conan_streams* s, int kind) {
  return guarded([&] {
    if (!s) throw Error(CONAN_ERR_INVALID, "null streams");
    if (kind < 0 || kind > 3) throw Error(CONAN_ERR_INVALID, "conan_streams_test_fault: kind 0 (off), 1 (decoder megakernel barrier), 2 (Emformer cluster exchange) or 3 (pair kernel flags)");
    s->test_fault = kind;
  });
}

int conan_streams_output_fence(conan_streams* s, void* fence_stream) {
  return guarded([&] {
    if (!s) throw Error(CONAN_ERR_INVALID, "null streams");
    s->fence_stream = (hipStream_t)fence_stream; s->fence_event = nullptr; s->fence_set = true;
  });
}

int conan_streams_output_fence_event(conan_streams* s, void* event) {
  return guarded([&] {
    if (!s || !event) throw Error(CONAN_ERR_INVALID, "null argument");
    s->fence_event = (hipEvent_t)event; s->fence_stream = nullptr; s->fence_set = true;
  });
}

int conan_streams_join(conan_streams* s, void* stream) {
  return guarded([&] {
    if (!s) throw Error(CONAN_ERR_INVALID, "null streams");
    HIP_CHECK(hipSetDevice(s->ctx->device)); s->check_fault();
    s->join((hipStream_t)stream);
  });
}

int conan_wav2mel(conan_ctx* ctx, const conan_mel_cfg* cfg, const float* wav_dev, int n, int samples, float* mel_out_dev,
                  int32_t* frames_out, void* stream) {
  return guarded([&] {
    if (!ctx || !cfg || !wav_dev || !mel_out_dev) throw Error(CONAN_ERR_INVALID, "null argument");
    if (!ctx->finalized) throw Error(CONAN_ERR_STATE, "conan_ctx_finalize must run before conan_wav2mel");
    HIP_CHECK(hipSetDevice(ctx->device));
    ctx->wav2mel(*cfg, wav_dev, n, samples, mel_out_dev, (hipStream_t)stream);
    if (frames_out) *frames_out = conan_mel_frames(*cfg, samples);
  });
}

int conan_profile_begin(conan_streams* s) {
  return guarded([&] {
    if (!s) throw Error(CONAN_ERR_INVALID, "null streams");
    s->prof_on = true; s->prof_used = 0; s->prof_flops = 0.0; s->prof_launches = 0; s->prof_rec.clear(); s->prof_kernels.clear();
  });
}

int conan_profile_end(conan_streams* s, double* conv_ms, double* conv_flops, int64_t* conv_launches) {
  return guarded([&] {
    if (!s) throw Error(CONAN_ERR_INVALID, "null streams");
    s->prof_on = false;
    double ms = 0.0;
    for (size_t i = 0; i < s->prof_used; ++i) {
      HIP_CHECK(hipEventSynchronize(s->prof_ev[i].second));
      float t = 0.f;
      HIP_CHECK(hipEventElapsedTime(&t, s->prof_ev[i].first, s->prof_ev[i].second));
      ms += t;
      const auto& r = s->prof_rec[i];
      bool found = false;
      for (auto& k : s->prof_kernels) if (k.name == r.name) { k.ms += t; k.flops += r.flops; k.n += 1; found = true; break; }
      if (!found) s->prof_kernels.push_back({r.name, (double)t, r.flops, 1});
    }
    if (conv_ms) *conv_ms = ms;
    if (conv_flops) *conv_flops = s->prof_flops;
    if (conv_launches) *conv_launches = s->prof_launches;
  });
}

int conan_profile_kernel(conan_streams* s, int index, char* name, int name_cap, double* ms, double* flops, int64_t* launches) {
  int found = 0;
  int rc = guarded([&] {
    if (!s) throw Error(CONAN_ERR_INVALID, "null streams");
    if (index < 0 || index >= (int)s->prof_kernels.size()) return;
    const auto& k = s->prof_kernels[index];
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", k.name.c_str());
    if (ms) *ms = k.ms;
    if (flops) *flops = k.flops;
    if (launches) *launches = k.n;
    found = 1;
  });
  return rc != CONAN_OK ? rc : found;
}

int conan_profile_mark(conan_streams* s, void* stream) {
  return guarded([&] {
    if (!s) throw Error(CONAN_ERR_INVALID, "null streams");
    HIP_CHECK(hipSetDevice(s->ctx->device)); s->check_fault();
    cnk::launch_profile_mark((hipStream_t)stream);
  });
}

int conan_hop_size(const conan_ctx* ctx) { return ctx ? ctx->hop : 0; }
int64_t conan_ctx_weight_bytes(const conan_ctx* ctx) { return ctx ? ctx->weight_bytes : 0; }
int64_t conan_streams_state_bytes(const conan_streams* s) { return s ? s->state_bytes : 0; }

}  // extern "C"
