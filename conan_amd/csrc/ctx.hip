// conan_ctx: weight ingestion (state_dict keys of the reference), weight-norm folding and repacking
// into the conv_mfma layout.  Host logic only; runs once at start-up.
#include "host_common.h"

namespace ch {
void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) throw Error(CONAN_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
}  // namespace ch

using ch::Error;
using ch::HostTensor;
using ch::PackedConv;

conan_ctx::~conan_ctx() {
  for (void* p : allocs) (void)hipFree(p);
}

float* conan_ctx::dev_alloc(size_t floats, bool zero) {
  void* p = nullptr;
  if (floats == 0) floats = 4;
  HIP_CHECK(hipMalloc(&p, floats * sizeof(float)));
  if (zero) HIP_CHECK(hipMemset(p, 0, floats * sizeof(float)));
  allocs.push_back(p);
  return (float*)p;
}

float* conan_ctx::upload(const std::vector<float>& v) {
  float* d = dev_alloc(v.size(), false);
  HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  weight_bytes += (int64_t)v.size() * 4;
  return d;
}

const HostTensor& conan_ctx::get(const std::string& key) const {
  auto it = raw.find(key);
  if (it == raw.end()) throw Error(CONAN_ERR_MISSING, "state_dict tensor not loaded: " + key);
  return it->second;
}

const PackedConv& conan_ctx::conv(const std::string& name) const {
  auto it = convs.find(name);
  if (it == convs.end()) throw Error(CONAN_ERR_STATE, "packed conv missing: " + name);
  return it->second;
}

float* conan_ctx::vec_or_null(const std::string& name) const {
  auto it = vecs.find(name);
  return it == vecs.end() ? nullptr : it->second;
}

float* conan_ctx::vec(const std::string& name) const {
  auto it = vecs.find(name);
  if (it == vecs.end()) throw Error(CONAN_ERR_STATE, "device vector missing: " + name);
  return it->second;
}

// W is PyTorch Conv1d layout [Cout][Cin][k] (Linear: k = 1).  Packed: [k][Cin_alloc/4][Cout_pad][4] with Cin_alloc =
// Cin rounded up to 128 (zero rows), so a 128-deep K-step may over-read past Cin_pad (= Cin rounded up to 32).
// shuffle_r > 1: output channel c*r + j of the reference (hifigan_causal.py:186-188) is stored at
// packed column j*(Cout/r) + c, which makes the pixel shuffle a plain row remap of the store.
void conan_ctx::pack_conv(const std::string& name, const std::vector<float>& W, const float* bias, int Cout, int Cin,
                          int k, int shuffle_r) {
  if (Cin % 4 != 0) throw Error(CONAN_ERR_UNSUPPORTED, "conv input channels must be a multiple of 4: " + name);
  PackedConv pc;
  pc.Cin = Cin; pc.Cout = Cout; pc.k = k; pc.shuffle_r = shuffle_r;
  pc.Cin_pad = ch::round_up(Cin, 32);
  pc.Cin_alloc = ch::round_up(Cin, 128);
  pc.Cout_pad = ch::round_up(Cout, 64);
  const int Cq = Cout / shuffle_r;
  std::vector<float> P((size_t)k * (pc.Cin_alloc / 4) * pc.Cout_pad * 4, 0.f);
  std::vector<float> B((size_t)pc.Cout_pad, 0.f);
  for (int co = 0; co < Cout; ++co) {
    int col = co;
    if (shuffle_r > 1) { int c = co / shuffle_r, j = co % shuffle_r; col = j * Cq + c; }
    if (bias) B[col] = bias[co];
    for (int ci = 0; ci < Cin; ++ci)
      for (int j = 0; j < k; ++j)
        P[((((size_t)(col / 64) * k + j) * (pc.Cin_alloc / 4) + ci / 4) * 64 + (col % 64)) * 4 + (ci & 3)] = W[((size_t)co * Cin + ci) * k + j];
  }
  pc.w = upload(P);
  pc.bias = upload(B);
  // bf16 limbs of the same (column-permuted) weights for conv_limb.hip: the vocoder's upsamplers and ResBlock convs
  if (name.rfind("voc.ups.", 0) == 0 || name.rfind("voc.rb.", 0) == 0) {
    if (Cin % 32 == 0) {
      auto rne = [](float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); };
      auto val = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
      const int NCT = ch::round_up(Cout, 16) / 16, NCB = Cin / 32;
      std::vector<uint16_t> lim((size_t)NCT * NCB * k * 3 * 512 + 4096, 0);
      for (int co = 0; co < Cout; ++co) {
        int col = co;
        if (shuffle_r > 1) { int c = co / shuffle_r, j = co % shuffle_r; col = j * Cq + c; }
        const int ct = col / 16, ln = col % 16;
        for (int ci = 0; ci < Cin; ++ci) {
          const int cb = ci / 32, lane = ln + 16 * ((ci % 32) / 8), e = ci % 8;
          for (int j = 0; j < k; ++j) {
            const float w = W[((size_t)co * Cin + ci) * k + j];
            const uint16_t h = rne(w); const float r1 = w - val(h);
            const uint16_t m = rne(r1); const float r2 = r1 - val(m);
            const uint16_t l = rne(r2);
            const size_t base = ((((size_t)ct * NCB + cb) * k + j) * 3) * 512 + (size_t)lane * 8 + e;
            lim[base] = h; lim[base + 512] = m; lim[base + 1024] = l;
          }
        }
      }
      std::vector<float> bits(lim.size() / 2);
      memcpy(bits.data(), lim.data(), bits.size() * 4);
      pc.wl = upload(bits);
      has_limb_weights = true;
    }
  }
  convs[name] = pc;
}

// Fragment-major copy of an already packed conv for rowconv.hip: per 16-column tile, per tap (k taps + one zero tap),
// per 16-deep K group one 1 KiB MFMA B operand (lane (g, n): W[tap][cin = 16 q + 4 g + s][cout = 16 ct + n], s = 0..3);
// columns padded to a multiple of 64 (256 for the wide layers, which run 4 column tiles per wave).
void conan_ctx::add_rowconv_weights(const std::string& name, const std::vector<float>& W) {
  PackedConv& pc = convs.at(name);
  const int Cin = pc.Cin, Cout = pc.Cout, k = pc.k;
  if (Cin % 64 || pc.shuffle_r != 1 || (Cin > 512 && !(k == 1 && Cin % 512 == 0 && Cin <= 4096))) return;   // (wide inputs: 1x1 layers only, rowconv.hip chunks their window)
  const int KQ = Cin / 16;
  if (KQ & (KQ - 1)) return;
  const int cpad = ch::round_up(Cout, Cout >= 1024 ? 256 : 64), NCT = cpad / 16;
  // (the kernels' weight rings run up to 8 groups ahead of the group they consume: a column tile's zero tap covers that
  // where KQ >= 8; the 8 zero groups behind the LAST column tile keep narrower layers - Cin = 64: KQ = 4 - in bounds too)
  std::vector<float> out((size_t)NCT * (k + 1) * KQ * 256 + 8 * 256, 0.f);
  for (int ct = 0; ct < NCT; ++ct)
    for (int j = 0; j < k; ++j)
      for (int q = 0; q < KQ; ++q)
        for (int lane = 0; lane < 64; ++lane)
          for (int s = 0; s < 4; ++s) {
            const int ci = q * 16 + 4 * (lane >> 4) + s, co = ct * 16 + (lane & 15);
            if (co < Cout) out[(((size_t)ct * (k + 1) + j) * KQ + q) * 256 + lane * 4 + s] = W[((size_t)co * Cin + ci) * k + j];
          }
  pc.wf = upload(out);
  pc.wf_cout_pad = cpad;
}

void conan_ctx::pack_from_keys(const std::string& name, const std::string& wkey, const std::string& bkey, int shuffle_r) {
  const HostTensor& w = get(wkey);
  if (w.shape.size() != 2 && w.shape.size() != 3) throw Error(CONAN_ERR_SHAPE, "bad weight rank: " + wkey);
  int Cout = (int)w.shape[0], Cin = (int)w.shape[1], k = w.shape.size() == 3 ? (int)w.shape[2] : 1;
  const float* b = nullptr;
  if (!bkey.empty()) { const HostTensor& bt = get(bkey); if (bt.numel() != Cout) throw Error(CONAN_ERR_SHAPE, "bad bias: " + bkey); b = bt.data.data(); }
  pack_conv(name, w.data, b, Cout, Cin, k, shuffle_r);
  // the frame-rate layers of the decoder step also get the rowconv layout
  if (name.rfind("conan.dec.", 0) == 0 || name.rfind("conan.uv.", 0) == 0 || name == "conan.content_proj" || name == "conan.mel_out" ||
      (name.rfind("conan.align.", 0) == 0 && name.find(".kv") == std::string::npos))
    add_rowconv_weights(name, w.data);
}

// weight_norm fold: w = g * v / ||v||_2 per output channel (torch._weight_norm(v, g, 0));
// accepts an already-folded '<prefix>.weight' (remove_weight_norm checkpoints, hifigan_causal.py:335).
void conan_ctx::fold_weightnorm(const std::string& prefix, std::vector<float>& W, std::vector<float>& bias, int& Cout, int& Cin, int& k) const {
  if (has(prefix + ".weight")) {
    const HostTensor& w = get(prefix + ".weight");
    Cout = (int)w.shape[0]; Cin = (int)w.shape[1]; k = (int)w.shape[2];
    W = w.data; bias = get(prefix + ".bias").data;
    return;
  }
  const HostTensor& v = get(prefix + ".weight_v");
  const HostTensor& g = get(prefix + ".weight_g");
  const HostTensor& b = get(prefix + ".bias");
  Cout = (int)v.shape[0]; Cin = (int)v.shape[1]; k = (int)v.shape[2];
  if (g.numel() != Cout || b.numel() != Cout) throw Error(CONAN_ERR_SHAPE, "bad weight_g/bias: " + prefix);
  W.resize(v.data.size());
  bias = b.data;
  const size_t per = (size_t)Cin * k;
  for (int co = 0; co < Cout; ++co) {
    // ||v|| accumulated in double and rounded once: within 1 ulp of torch._weight_norm's fp32 norm
    double s = 0.0;
    for (size_t e = 0; e < per; ++e) { double x = v.data[co * per + e]; s += x * x; }
    const float nrm = (float)std::sqrt(s);
    const float gg = g.data[co];
    for (size_t e = 0; e < per; ++e) W[co * per + e] = v.data[co * per + e] * (gg / nrm);
  }
}

void conan_ctx::pack_weightnorm(const std::string& name, const std::string& prefix, int shuffle_r, bool rowconv) {
  std::vector<float> W, b;
  int Cout, Cin, k;
  fold_weightnorm(prefix, W, b, Cout, Cin, k);
  pack_conv(name, W, b.data(), Cout, Cin, k, shuffle_r);
  if (rowconv) add_rowconv_weights(name, W);
}

// Weights of a C -> C conv for resblock_fused.hip: per 16-column tile, per tap (k taps + one zero tap), per 16-deep K
// group one 1 KiB MFMA B operand: lane (g = lane >> 4, n = lane & 15) holds W[tap][cin = 16 q + 4 g + s][cout = 16 ct + n],
// s = 0..3.  vecs[name + ".w"], vecs[name + ".b"].
void conan_ctx::pack_fragments(const std::string& name, const std::string& prefix) {
  std::vector<float> W, b;
  int Cout, Cin, k;
  fold_weightnorm(prefix, W, b, Cout, Cin, k);
  if (Cout != Cin || Cin % 16) throw Error(CONAN_ERR_SHAPE, "fused resblock conv must be C -> C with C a multiple of 16: " + prefix);
  const int C = Cin, KQ = C / 16, NCT = C / 16;
  std::vector<float> out((size_t)NCT * (k + 1) * KQ * 256, 0.f);
  for (int ct = 0; ct < NCT; ++ct)
    for (int j = 0; j < k; ++j)
      for (int q = 0; q < KQ; ++q)
        for (int lane = 0; lane < 64; ++lane)
          for (int s = 0; s < 4; ++s) {
            const int ci = q * 16 + 4 * (lane >> 4) + s, co = ct * 16 + (lane & 15);
            out[(((size_t)ct * (k + 1) + j) * KQ + q) * 256 + lane * 4 + s] = W[((size_t)co * C + ci) * k + j];
          }
  vecs[name + ".w"] = upload(out);
  vecs[name + ".b"] = upload(b);
  // The same weights as three bf16 limbs each, w = h + m + l (round-to-nearest-even; the subtractions are exact), for the
  // bf16-MFMA form of the pass (resblock_limb.hip): [ct][k + 1 taps][C/32 K blocks][3 limbs][64 lanes][8 elements], lane =
  // (output channel ct*16 + (lane & 15), input channels kb*32 + 8*(lane >> 4) + e).  Stored behind a float pointer (bits only).
  if (C % 32 == 0 && cnk::resblock_limb_supported(C, k, 0)) {
    auto rne = [](float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); };
    auto val = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
    const int KB = C / 32;
    std::vector<uint16_t> lim((size_t)NCT * (k + 1) * KB * 3 * 512 + 4096, 0);      // + a block of slack: the ring prefetches past the last tap
    for (int ct = 0; ct < NCT; ++ct)
      for (int j = 0; j < k; ++j)
        for (int q = 0; q < KB; ++q)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
              const int ci = q * 32 + 8 * (lane >> 4) + e, co = ct * 16 + (lane & 15);
              const float w = W[((size_t)co * C + ci) * k + j];
              const uint16_t h = rne(w); const float r1 = w - val(h);
              const uint16_t m = rne(r1); const float r2 = r1 - val(m);
              const uint16_t l = rne(r2);
              const size_t base = ((((size_t)ct * (k + 1) + j) * KB + q) * 3) * 512 + lane * 8 + e;
              lim[base] = h; lim[base + 512] = m; lim[base + 1024] = l;
            }
    std::vector<float> bits(lim.size() / 2);
    memcpy(bits.data(), lim.data(), bits.size() * 4);
    vecs[name + ".wl"] = upload(bits);
    has_limb_weights = true;
  }
}

void conan_ctx::upload_vec(const std::string& name, const std::string& key) { vecs[name] = upload(get(key).data); }

void conan_ctx::finalize_hifigan() {
  const conan_cfg& c = cfg;
  const std::string P = "hifigan.";
  pack_weightnorm("voc.conv_pre", P + "conv_pre.conv");
  int ridx = 0;
  for (int i = 0; i < c.voc_num_ups; ++i) {
    const std::string up = P + "ups." + std::to_string(i) + ".conv.conv";
    if (c.voc_upsample == 0) {
      pack_weightnorm("voc.ups." + std::to_string(i), up, c.voc_up_rates[i]);
    } else if (c.voc_upsample == 2) {
      // CausalUpsampleBlock1 (hifigan_causal.py:60-145): left pad P = k/2 - 1, ConvTranspose1d, drop P*s + k - 1 samples:
      //   y[q*s + r][co] = b[co] + sum_m sum_ci x[q + m][ci] * w[ci][co][r + k - 1 - m*s],   0 <= r + k - 1 - m*s < k,
      // i.e. m = 0 .. M = (k + s - 2) / s input frames AHEAD of q and none behind it (x beyond the last frame = 0).
      // Packed as an M + 1 tap conv to Cout*s channels in pixel-shuffle form; the launch reads rows t .. t + M
      // (pad_left = 0, streams.hip) instead of t - M .. t.  weight_norm of a ConvTranspose1d normalises over dim 0 of the
      // [Cin, Cout, k] weight, the INPUT channel.
      const std::string dc = P + "ups." + std::to_string(i) + ".deconv";
      const bool folded = has(dc + ".weight");
      const HostTensor& v = get(dc + (folded ? ".weight" : ".weight_v"));
      const HostTensor& bias = get(dc + ".bias");
      const int Cin = (int)v.shape[0], Cout = (int)v.shape[1], k = (int)v.shape[2];
      const int sr = c.voc_up_rates[i];
      if (k != c.voc_up_kernels[i] || k % 2 || sr < 2 || bias.numel() != Cout) throw Error(CONAN_ERR_SHAPE, "bad transposed upsampler: " + dc);
      std::vector<float> W(v.data);
      if (!folded) {
        const HostTensor& g = get(dc + ".weight_g");
        if (g.numel() != Cin) throw Error(CONAN_ERR_SHAPE, "bad weight_g: " + dc);
        const size_t per = (size_t)Cout * k;
        for (int ci = 0; ci < Cin; ++ci) {
          double ss = 0.0;
          for (size_t e = 0; e < per; ++e) { double x = v.data[ci * per + e]; ss += x * x; }
          const float nrm = (float)std::sqrt(ss), gg = g.data[ci];
          for (size_t e = 0; e < per; ++e) W[ci * per + e] = v.data[ci * per + e] * (gg / nrm);
        }
      }
      const int D = (k + sr - 2) / sr + 1;
      std::vector<float> Wp((size_t)Cout * sr * Cin * D, 0.f), bp((size_t)Cout * sr);
      for (int co = 0; co < Cout; ++co)
        for (int r = 0; r < sr; ++r) {
          bp[(size_t)co * sr + r] = bias.data[co];
          for (int m = 0; m < D; ++m) {
            const int idx = r + k - 1 - m * sr;
            if (idx < 0 || idx >= k) continue;
            for (int ci = 0; ci < Cin; ++ci)     // tap m reads x[t + m]
              Wp[(((size_t)co * sr + r) * Cin + ci) * D + m] = W[((size_t)ci * Cout + co) * k + idx];
          }
        }
      pack_conv("voc.ups." + std::to_string(i), Wp, bp.data(), Cout * sr, Cin, D, sr);
    } else {
      // CausalUpsampleBlock2 (zero insertion + causal conv k, hifigan_causal.py:151-165) as a polyphase conv over the
      // *input* rate in pixel-shuffle form: output sample t*s+j = sum_d w[(k-1) - j - s*d] . x[t-d], d = 0 .. D-1 with
      // D = (k-1)/s + 1; phases whose index leaves [0, k) get zero taps.  Same kernel, same free shuffle as 'shuffle'.
      std::vector<float> W, bias;
      int Cout, Cin, k;
      fold_weightnorm(up, W, bias, Cout, Cin, k);
      const int sr = c.voc_up_rates[i], D = (k - 1) / sr + 1;
      std::vector<float> Wp((size_t)Cout * sr * Cin * D, 0.f), bp((size_t)Cout * sr);
      for (int co = 0; co < Cout; ++co)
        for (int j = 0; j < sr; ++j) {
          bp[(size_t)co * sr + j] = bias[co];
          for (int d = 0; d < D; ++d) {
            const int idx = (k - 1) - j - sr * d;
            if (idx < 0 || idx >= k) continue;
            for (int ci = 0; ci < Cin; ++ci)     // causal-conv tap q = D-1-d reads x[t-d]
              Wp[(((size_t)co * sr + j) * Cin + ci) * D + (D - 1 - d)] = W[((size_t)co * Cin + ci) * k + idx];
          }
        }
      pack_conv("voc.ups." + std::to_string(i), Wp, bp.data(), Cout * sr, Cin, D, sr);
    }
    for (int b = 0; b < c.voc_num_resblocks; ++b, ++ridx)
      for (int d = 0; d < c.voc_rb_num_dil; ++d) {
        std::string rb = "resblocks." + std::to_string(ridx);
        if (c.voc_resblock == 2) {
          pack_weightnorm("voc.rb." + std::to_string(ridx) + ".c." + std::to_string(d), P + rb + ".convs." + std::to_string(d) + ".conv");
        } else {
          const int Cs = c.voc_initial_channel >> (i + 1);
          const bool rc = false;      // (the rowconv layout of these convs was an experiment: nothing launches it)
          pack_weightnorm("voc.rb." + std::to_string(ridx) + ".c1." + std::to_string(d), P + rb + ".convs1." + std::to_string(d) + ".conv", 1, rc);
          pack_weightnorm("voc.rb." + std::to_string(ridx) + ".c2." + std::to_string(d), P + rb + ".convs2." + std::to_string(d) + ".conv", 1, rc);
          if (cnk::resblock_fused_supported(Cs, c.voc_rb_kernels[b], (c.voc_rb_kernels[b] - 1) * c.voc_rb_dilations[b][d]) ||
              cnk::resblock_pair_supported(Cs, c.voc_rb_kernels[b], (c.voc_rb_kernels[b] - 1) * c.voc_rb_dilations[b][d], 32)) {
            pack_fragments("voc.rbf." + std::to_string(ridx) + ".c1." + std::to_string(d), P + rb + ".convs1." + std::to_string(d) + ".conv");
            pack_fragments("voc.rbf." + std::to_string(ridx) + ".c2." + std::to_string(d), P + rb + ".convs2." + std::to_string(d) + ".conv");
          }
        }
      }
  }
  pack_weightnorm("voc.conv_post", P + "conv_post.conv");
  {  // conv_post also as a [k][C] vector for the VALU kernel (Cout == 1)
    std::vector<float> W, b;
    int Cout, Cin, k;
    fold_weightnorm(P + "conv_post.conv", W, b, Cout, Cin, k);
    if (Cout != 1) throw Error(CONAN_ERR_SHAPE, "conv_post must have one output channel");
    std::vector<float> T((size_t)k * Cin);
    for (int ci = 0; ci < Cin; ++ci) for (int j = 0; j < k; ++j) T[(size_t)j * Cin + ci] = W[(size_t)ci * k + j];
    vecs["voc.conv_post.w"] = upload(T);
    scalars["voc.conv_post.b"] = b[0];
    scalars["voc.conv_post.k"] = (float)k;
  }
}

void conan_ctx::finalize_emformer() {
  const conan_cfg& c = cfg;
  for (int l = 0; l < c.emf_layers; ++l) {
    std::string p = "emformer.emformer.emformer_layers." + std::to_string(l);
    std::string n = "emf." + std::to_string(l);
    pack_from_keys(n + ".q", p + ".attention.emb_to_query.weight", p + ".attention.emb_to_query.bias");
    pack_from_keys(n + ".kv", p + ".attention.emb_to_key_value.weight", p + ".attention.emb_to_key_value.bias");
    pack_from_keys(n + ".out", p + ".attention.out_proj.weight", p + ".attention.out_proj.bias");
    pack_from_keys(n + ".ff1", p + ".pos_ff.1.weight", p + ".pos_ff.1.bias");
    pack_from_keys(n + ".ff2", p + ".pos_ff.4.weight", p + ".pos_ff.4.bias");
    upload_vec(n + ".ln_in.g", p + ".layer_norm_input.weight");
    upload_vec(n + ".ln_in.b", p + ".layer_norm_input.bias");
    upload_vec(n + ".ln_ff.g", p + ".pos_ff.0.weight");
    upload_vec(n + ".ln_ff.b", p + ".pos_ff.0.bias");
    upload_vec(n + ".ln_out.g", p + ".layer_norm_output.weight");
    upload_vec(n + ".ln_out.b", p + ".layer_norm_output.bias");
  }
  // mode == 'both' checkpoints carry proj1 (80 -> 100) / proj2 (80 -> 768) heads; the streaming loop reads proj1
  // (modules/Emformer/emformer.py:28-30, inference/Conan.py:117-118)
  const std::string pj = has("emformer.proj1.weight") ? "emformer.proj1" : "emformer.proj";
  if (c.emf_output_dim != c.emf_input_dim) {
    if (get(pj + ".weight").shape.size() != 2 || get(pj + ".weight").shape[0] != c.emf_output_dim)
      throw Error(CONAN_ERR_SHAPE, "Emformer projection rows != emf_output_dim: " + pj);
    pack_from_keys("emf.proj", pj + ".weight", pj + ".bias");
  }
  // every output head the checkpoint carries, by name, for conan_emformer_project (EmformerDistillModel.inference projects
  // the concatenated features, modules/Emformer/emformer.py:95-97)
  for (const char* h : {"proj", "proj1", "proj2"}) {
    const std::string k = std::string("emformer.") + h;
    if (has(k + ".weight") && get(k + ".weight").shape.size() == 2 && get(k + ".weight").shape[1] == c.emf_input_dim)
      pack_from_keys(std::string("emf.head.") + h, k + ".weight", k + ".bias");
  }
  // Second copy of the Linear weights for the fused step (emformer_fused.hip), fragment-major: fragment (ntile, kq)
  // is the 64-lane x float4 MFMA B operand {W[ntile*16 + (lane&15)][kq*16 + (lane>>4)*4 + e]} stored as 1 KiB, and
  // fragments are ordered the way a wave consumes them, so each wave streams its weights sequentially.
  const int D = c.emf_input_dim, F = c.emf_ffn_dim;
  if (D % 16 || F % 64) return;   // the fused step does not cover such shapes (streams.hip falls back per op)
  auto frags = [&](const std::vector<float>& W, int N, int K, size_t nfr, auto index) {
    std::vector<float> out(nfr * 256, 0.f);
    const int nt = (N + 15) / 16, kqs = K / 16;
    for (int t = 0; t < nt; ++t)
      for (int kq = 0; kq < kqs; ++kq) {
        float* f = out.data() + index(t, kq) * 256;
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 4; ++e) {
            const int n = t * 16 + (lane & 15), k = kq * 16 + (lane >> 4) * 4 + e;
            f[lane * 4 + e] = n < N ? W[(size_t)n * K + k] : 0.f;
          }
      }
    return out;
  };
  const int KQ = D / 16;
  for (int l = 0; l < c.emf_layers; ++l) {
    std::string p = "emformer.emformer.emformer_layers." + std::to_string(l);
    std::string n = "emf." + std::to_string(l);
    std::vector<float> qkv = get(p + ".attention.emb_to_query.weight").data;          // [D][D] then [2D][D]
    const std::vector<float>& kv = get(p + ".attention.emb_to_key_value.weight").data;
    qkv.insert(qkv.end(), kv.begin(), kv.end());
    vecs[n + ".fqkv"] = upload(frags(qkv, 3 * D, D, (size_t)(3 * D / 16) * KQ, [&](int t, int kq) { return (size_t)t * KQ + kq; }));
    vecs[n + ".fo"] = upload(frags(get(p + ".attention.out_proj.weight").data, D, D, (size_t)KQ * KQ, [&](int t, int kq) { return (size_t)t * KQ + kq; }));
    // FF1: units of 4 tiles (64 hidden columns = one wave's share of a chunk), [unit][kq][tile in unit]
    vecs[n + ".f1"] = upload(frags(get(p + ".pos_ff.1.weight").data, F, D, (size_t)(F / 16) * KQ, [&](int t, int kq) { return ((size_t)(t / 4) * KQ + kq) * 4 + (t % 4); }));
    // FF2: k-group major, [kq][tile]
    vecs[n + ".f2"] = upload(frags(get(p + ".pos_ff.4.weight").data, D, F, (size_t)(F / 16) * KQ, [&](int t, int kq) { return (size_t)kq * KQ + t; }));
  }
  if (c.emf_output_dim != D) {
    const int nt = (c.emf_output_dim + 15) / 16;
    vecs["emf.fproj"] = upload(frags(get(pj + ".weight").data, c.emf_output_dim, D, (size_t)nt * KQ, [&](int t, int kq) { return (size_t)t * KQ + kq; }));
  }
}

void conan_ctx::finalize_conan() {
  const conan_cfg& c = cfg;
  const std::string P = "conan.";
  const int H = c.hidden_size;
  upload_vec("conan.content_embedding", P + "content_embedding.weight");
  upload_vec("conan.pitch_embed", P + "pitch_embed.weight");
  pack_from_keys("conan.content_proj", P + "content_proj.0.conv.weight", P + "content_proj.0.conv.bias");
  // aligner: in_proj rows [0,H) = Wq, [H,3H) = Wk|Wv
  for (int l = 0; l < 2; ++l) {
    std::string p = P + "align.layers." + std::to_string(l);
    std::string n = "conan.align." + std::to_string(l);
    const HostTensor& w = get(p + ".multihead_attn.in_proj_weight");
    const HostTensor& b = get(p + ".multihead_attn.in_proj_bias");
    if (w.shape.size() != 2 || w.shape[0] != 3 * H || w.shape[1] != H) throw Error(CONAN_ERR_SHAPE, "in_proj_weight");
    std::vector<float> wq(w.data.begin(), w.data.begin() + (size_t)H * H);
    std::vector<float> wkv(w.data.begin() + (size_t)H * H, w.data.end());
    pack_conv(n + ".q", wq, b.data.data(), H, H, 1);
    add_rowconv_weights(n + ".q", wq);
    pack_conv(n + ".kv", wkv, b.data.data() + H, 2 * H, H, 1);
    pack_from_keys(n + ".out", p + ".multihead_attn.out_proj.weight", p + ".multihead_attn.out_proj.bias");
    pack_from_keys(n + ".ff1", p + ".linear1.weight", p + ".linear1.bias");
    pack_from_keys(n + ".ff2", p + ".linear2.weight", p + ".linear2.bias");
    upload_vec(n + ".norm1.g", p + ".norm1.weight"); upload_vec(n + ".norm1.b", p + ".norm1.bias");
    upload_vec(n + ".norm2.g", p + ".norm2.weight"); upload_vec(n + ".norm2.b", p + ".norm2.bias");
  }
  // The widths the reference's constructors hard-code - uv predictor n_chans = 128 and 5 layers (Conan.py:106-113), aligner
  // dim_feedforward = 2048 and 2 layers (prosody_util.py:97,133, Conan.py:81) - are taken from the checkpoint's tensors, so a
  // checkpoint trained with other widths loads; what the kernels cannot run fails HERE with CONAN_ERR_SHAPE, not in a step.
  if (has(P + "align.layers.2.linear1.weight")) throw Error(CONAN_ERR_SHAPE, "ProsodyAligner with more than 2 layers (conan_decoder_taps carries two attention maps)");
  for (int l = 0; l < 2; ++l) {
    const ch::PackedConv &f1 = conv("conan.align." + std::to_string(l) + ".ff1"), &f2 = conv("conan.align." + std::to_string(l) + ".ff2");
    if (f1.Cin != H || f2.Cout != H || f2.Cin != f1.Cout || f1.k != 1 || f2.k != 1) throw Error(CONAN_ERR_SHAPE, "aligner feed-forward shapes (linear1 [F,H], linear2 [H,F])");
    if (l && f1.Cout != conv("conan.align.0.ff1").Cout) throw Error(CONAN_ERR_SHAPE, "aligner layers with different feed-forward widths");
  }
  scalars["conan.align.ffn"] = (float)conv("conan.align.0.ff1").Cout;
  // uv predictor: conv.i for as many layers as the checkpoint holds (nar_tts_modules.py:113-122)
  int n_uv = 0;
  while (has(P + "uv_predictor.conv." + std::to_string(n_uv) + ".0.conv.weight")) ++n_uv;
  if (n_uv < 1 || n_uv > 8) throw Error(CONAN_ERR_SHAPE, "uv_predictor: 1 to 8 conv layers expected");
  for (int i = 0; i < n_uv; ++i) {
    std::string p = P + "uv_predictor.conv." + std::to_string(i) + ".0.conv";
    pack_from_keys("conan.uv." + std::to_string(i), p + ".weight", p + ".bias");
    const ch::PackedConv& pc = conv("conan.uv." + std::to_string(i));
    if (pc.Cin != (i == 0 ? H : conv("conan.uv." + std::to_string(i - 1)).Cout) || pc.k != c.predictor_kernel)
      throw Error(CONAN_ERR_SHAPE, "uv_predictor.conv." + std::to_string(i) + ": input width / kernel size do not chain");
  }
  const int uvh = conv("conan.uv." + std::to_string(n_uv - 1)).Cout;
  if (uvh > 256 || get(P + "uv_predictor.post_ln.weight").numel() != uvh || get(P + "uv_predictor.linear.weight").numel() != 2 * (long long)uvh)
    throw Error(CONAN_ERR_SHAPE, "uv_predictor: post_ln / linear do not match the last conv's width (at most 256 channels)");
  scalars["conan.uv.n"] = (float)n_uv; scalars["conan.uv.hidden"] = (float)uvh;
  upload_vec("conan.uv.ln.g", P + "uv_predictor.post_ln.weight");
  upload_vec("conan.uv.ln.b", P + "uv_predictor.post_ln.bias");
  upload_vec("conan.uv.lin.w", P + "uv_predictor.linear.weight");
  upload_vec("conan.uv.lin.b", P + "uv_predictor.linear.bias");
  // decoder (CausalConvBlocks)
  auto conv_blocks = [&](const std::string& src, const std::string& dst, int nblocks, int nin, bool causal) {
    const char* ic = causal ? ".2" : ".1";
    const char* i1 = causal ? ".5" : ".4";
    for (int b = 0; b < nblocks; ++b)
      for (int j = 0; j < nin; ++j) {
        std::string p = src + ".res_blocks." + std::to_string(b) + ".blocks." + std::to_string(j);
        std::string n = dst + "." + std::to_string(b) + "." + std::to_string(j);
        upload_vec(n + ".ln.g", p + ".0.weight"); upload_vec(n + ".ln.b", p + ".0.bias");
        pack_from_keys(n + ".c1", p + ic + ".weight", p + ic + ".bias");
        pack_from_keys(n + ".c2", p + i1 + ".weight", p + i1 + ".bias");
      }
    upload_vec(dst + ".last.g", src + ".last_norm.weight"); upload_vec(dst + ".last.b", src + ".last_norm.bias");
    std::string pn = causal ? src + ".post_net1.1" : src + ".post_net1";
    pack_from_keys(dst + ".post", pn + ".weight", pn + ".bias");
  };
  conv_blocks(P + "decoder", "conan.dec", c.dec_num_blocks, c.dec_layers_in_block, true);
  pack_from_keys("conan.mel_out", P + "mel_out.weight", P + "mel_out.bias");
  // style pass
  pack_from_keys("conan.global_conv_in", P + "global_conv_in.weight", P + "global_conv_in.bias");
  conv_blocks(P + "global_encoder", "conan.genc", 5, 2, false);
  conv_blocks(P + "prosody_extractor.encoder", "conan.penc", 5, 2, false);
  for (int i = 0; i < 4; ++i) {
    pack_weightnorm("conan.wn.in." + std::to_string(i), P + "prosody_extractor.wavenet.in_layers." + std::to_string(i));
    pack_weightnorm("conan.wn.rs." + std::to_string(i), P + "prosody_extractor.wavenet.res_skip_layers." + std::to_string(i));
  }
  {  // the style pass works on the reference mel itself: WN(hidden = num_mels), encoder width num_mels (prosody_util.py:173-181)
    const int NM = c.num_mels;
    if (conv("conan.global_conv_in").Cin != NM || conv("conan.global_conv_in").Cout != H) throw Error(CONAN_ERR_SHAPE, "global_conv_in must be num_mels -> hidden_size");
    for (int i = 0; i < 4; ++i) {
      const ch::PackedConv &wi = conv("conan.wn.in." + std::to_string(i)), &wr = conv("conan.wn.rs." + std::to_string(i));
      if (wi.Cin != NM || wi.Cout != 2 * NM || wr.Cin != NM || wr.Cout != (i < 3 ? 2 * NM : NM)) throw Error(CONAN_ERR_SHAPE, "prosody_extractor.wavenet: WN(hidden = num_mels) expected");
    }
    if (conv("conan.penc.0.0.c1").Cin != NM) throw Error(CONAN_ERR_SHAPE, "prosody_extractor.encoder width must be num_mels");
  }
  {
    const HostTensor& e = get(P + "prosody_extractor.vqvae.embedding");
    if (e.shape.size() != 2 || e.shape[1] != H) throw Error(CONAN_ERR_SHAPE, "vqvae.embedding");
    const int M = (int)e.shape[0];
    vecs["conan.vq.emb"] = upload(e.data);
    pack_conv("conan.vq.dot", e.data, nullptr, M, H, 1);      // dots[s][j] = x . e_j
    std::vector<float> e2(M);
    for (int j = 0; j < M; ++j) { float s = 0.f; for (int d = 0; d < H; ++d) { float x = e.data[(size_t)j * H + d]; s += x * x; } e2[j] = s; }
    vecs["conan.vq.e2"] = upload(e2);
  }
  pack_from_keys("conan.l1", P + "l1.weight", P + "l1.bias");
  {
    // SinusoidalPositionalEmbedding.get_embedding (modules/commons/transformer.py:30-47), fp32 steps
    const int n = 2002, half = H / 2;
    std::vector<float> tab((size_t)n * H, 0.f);
    const float e = (float)(std::log(10000.0) / (half - 1));
    for (int p = 1; p < n; ++p)
      for (int d = 0; d < half; ++d) {
        float f = std::exp((float)d * -e);
        float a = (float)p * f;
        tab[(size_t)p * H + d] = std::sin(a);
        tab[(size_t)p * H + half + d] = std::cos(a);
      }
    vecs["conan.postable"] = upload(tab);
  }
}
