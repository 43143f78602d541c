// conan_streams: per-slot streaming state (activation rings with their left context, Emformer K/V
// rings, cached style pass) and the launch plans of the vocoder and Emformer steps.
#include "streams.h"


ConvArgs conan_streams::mk(const PackedConv& pc, const TRef& x, const TRef& y, int n, int T, const int* pos, int dil,
                           int pad_left) const {
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x;
  a.y = y; a.res = ch::null_ref(); a.m1 = ch::null_ref(); a.m2 = ch::null_ref();
  a.w = pc.w; a.bias = pc.bias; a.bvec = nullptr; a.slots = d_slots; a.pos = pos; a.lens = nullptr;
  a.Cin = pc.Cin; a.Cin_pad = pc.Cin_pad; a.Cin_alloc = pc.Cin_alloc; a.Cout = pc.Cout; a.Cout_pad = pc.Cout_pad;
  a.ktaps = pc.k; a.dil = dil; a.pad_left = pad_left < 0 ? (pc.k - 1) * dil : pad_left;
  a.T = T; a.n = n; a.in_act = cnk::ACT_NONE; a.in_slope = 0.f; a.out_act = cnk::ACT_NONE; a.out_scale = 1.f; a.out_slope = 0.f;
  a.shuffle_r = pc.shuffle_r;
  a.wl = reinterpret_cast<const unsigned short*>(pc.wl);
  if (x.C != pc.Cin && !(x.C > pc.Cin)) throw Error(CONAN_ERR_SHAPE, "conv input width mismatch");
  return a;
}

int conan_streams::pick_cfg(int M, int N, int nprob) const {
  // Measured on MI355X (tools/conv_bench.hip): 64x64 tiles (2 persistent blocks per CU) beat 128x64 (1 block per CU:
  // its 3-deep direct-to-LDS ring needs 92 KB) on every streaming layer; below one block per CU the small-M shapes
  // with intra-block split-K take over, and when even those cannot fill the chip the one with the most blocks
  // (those layers are latency-bound).
  const int wide[] = {cnk::CFG_64x64, cnk::CFG_32x64_K2, cnk::CFG_32x32_K4};
  const int narrow[] = {cnk::CFG_128x32, cnk::CFG_64x32_K2, cnk::CFG_32x32_K4};
  const int* order = N <= 32 ? narrow : wide;
  const int cnt = 3;
  long long need = ctx->num_cu;
  int best = order[cnt - 1];
  for (int k = 0; k < cnt; ++k) {
    int c = order[k];
    long long blocks = (long long)((M + cnk::conv_cfg_tm(c) - 1) / cnk::conv_cfg_tm(c)) * ((N + cnk::conv_cfg_tn(c) - 1) / cnk::conv_cfg_tn(c)) * nprob;
    if (blocks >= need) {
      best = c;
      // too few 64x64 tiles for two blocks per CU (stage-1 resblocks at 64 streams: 384 tiles): the K-step-64 build runs
      // one persistent block per CU over a balanced tile list instead of leaving a third of the CUs half empty
      // - for the grouped launches, whose tiles of three costs the list balances.  A single problem with 1.25 equal tiles
      // per CU (ups.1: 320) needs two rounds either way; there the K-step-32 build with its 320 independently dispatched
      // blocks is faster alone (94 against 109 us) and, above all, in the pipelined step, where a CU that another
      // stream's block holds for a while delays a whole static list (step 1.856 -> 1.805 ms)
      if (c == cnk::CFG_64x64 && blocks < 2 * need && nprob > 1) best = cnk::CFG_64x64_KS64;
      break;
    }
  }
  return best;
}

void conan_streams::launch_group(const ConvGroup& gin, int nprob, int cfg, hipStream_t st) {
  if (mega_rec) { mega_rec_ok = false; return; }      // (a conv_mfma layer is not a megakernel operator: that step keeps its separate launches)
  for (int p = 0; p < nprob; ++p) {
    // both conv kernels address activations with 32-bit offsets from the tensor base (bytes in conv_mfma's direct-to-LDS loader,
    // floats in conv_limb's window staging)
    const ConvArgs& a = gin.p[p];
    if ((long long)(a.x.mode == 0 ? max_slots : a.n) * a.x.slot_stride * 4 >= (1ll << 32))
      throw Error(CONAN_ERR_UNSUPPORTED, "activation tensor of 4 GiB or more: lower max_slots");
    if (a.y2_base && ((a.Cout & 3) || (a.y.C & 3) || ((a.Cout / a.shuffle_r) & 3))) throw Error(CONAN_ERR_UNSUPPORTED, "activated twin output needs channel counts that are multiples of 4");
  }
  // few rows per slot and a long K (ups.0: 4 rows per stream, K = 8192): the split-K limb GEMM of conv_tall.hip
  if (rb_limb && nprob == 1 && dev("NO_TALL") == nullptr) {
    const int cus = std::max(8, ctx->num_cu - (ws_index(st) == 1 ? reserve_cus : 0));
    const int wsi = ws_index(st);
    cnk::ConvTallArgs ta;
    if (cnk::conv_tall_plan(gin.p[0], cus, fixed_plan ? plan_n(gin.p[0].n) : 0, &ta, sk_slab_floats, sk_max_tiles, dev("TALL_MAXT") ? atoi(dev("TALL_MAXT")) : 32)) {
      ta.slab = sk_slab[wsi]; ta.counters = sk_counters[wsi];
      const ConvArgs& a = gin.p[0];
      profiled("cnk::conv_tall_kernel", 2.0 * (double)a.n * a.T * a.Cout * a.ktaps * a.Cin, st, [&] { cnk::launch_conv_tall(ta, st); });
      return;
    }
  }
  // the bf16-limb form (conv_limb.hip) where the weights were packed for it and a tile shape fits all problems of the group
  if (rb_limb && nprob >= 1 && nprob <= 3) {
    bool ok = true;
    for (int p = 0; p < nprob; ++p) ok = ok && cnk::conv_limb_supported(gin.p[p]) && gin.p[p].n == gin.p[0].n && gin.p[p].T == gin.p[0].T && gin.p[p].Cout == gin.p[0].Cout;
    const int cus = std::max(8, ctx->num_cu - (ws_index(st) == 1 ? reserve_cus : 0));
    // (a single problem may split the K range of its tail tiles - conv_limb.hip; not in a fixed-plan stream-set: which tiles are the
    // tail follows a slot's position in the active list)
    const int shape = ok ? cnk::conv_limb_shape(gin.p, nprob, cus, fixed_plan ? plan_n(gin.p[0].n) : 0, !fixed_plan) : -1;
    if (shape >= 0) {
      cnk::ConvLimbGroup lg; memset(&lg, 0, sizeof(lg));
      for (int p = 0; p < nprob; ++p) lg.p[p] = gin.p[p];
      lg.nprob = nprob;
      if (!fixed_plan && nprob == 1) {
        const int wsi = ws_index(st);
        lg.slab = sk_slab[wsi]; lg.counters = sk_counters[wsi]; lg.slab_floats = sk_slab_floats; lg.max_counters = sk_max_tiles;
      }
      double fl = 0.0;
      for (int p = 0; p < nprob; ++p) fl += 2.0 * (double)lg.p[p].n * lg.p[p].T * lg.p[p].Cout * lg.p[p].ktaps * lg.p[p].Cin;
      const bool tail = lg.slab && shape == 0 && cnk::conv_limb_tail_slices(lg.p[0], shape, cus) >= 2;
      profiled(tail ? "cnk::conv_limb_sk_kernel<4, 1, 1, 4>" : cnk::conv_limb_name(shape), fl, st, [&] {
        if (!cnk::launch_conv_limb(lg, shape, cus, st)) throw Error(CONAN_ERR_HIP, "conv_limb launch failed");
      });
      return;
    }
  }
  ConvGroup g = gin;
  for (int p = 0; p < nprob; ++p) {
    // layers that touch no per-slot ring need neither the slot table nor the position counters: dropping them
    // removes two dependent global loads from the prologue and from the epilogue of these latency-bound launches
    ConvArgs& a = g.p[p];
    bool ring = a.y.mode == 0 || (a.has_res && a.res.mode == 0) || (a.has_m1 && a.m1.mode == 0) || (a.has_m2 && a.m2.mode == 0) || a.bvec != nullptr;
    ring = ring || a.x.mode == 0;
    if (!ring) { a.slots = nullptr; a.pos = nullptr; }
  }
  // inter-block split-K for launches that cannot fill the chip with tiles but have a long K loop
  const int wsi = ws_index(st);
  g.slab = sk_slab[wsi]; g.counters = sk_counters[wsi]; g.ksplit = 1; g.split_from = 0; g.fenced = fenced ? 1 : 0;
  const int cus_all = std::max(8, ctx->num_cu - (wsi == 1 ? reserve_cus : 0));
  if (cnk::conv_cfg_tm(cfg) != 32 && nprob == 1 && cnk::conv_cfg_splitk(cfg)) {
    // Streaming shapes whose tile count is not a multiple of the CU count (ups[1]: 320 tiles of 64 x 64 on 256 CUs): the
    // last `rem` tiles are cut into S K-slices each, rem * S <= CUs, so that every CU gets the same 1 + 1/S (2 + 1/S ...)
    // tiles of MFMA work instead of some CUs getting one tile more than the others - the tail of a stream-K schedule,
    // with the fence-free partial-tile hand-off of the small-M shapes.
    static const bool off = ch::dev_getenv("CONAN_NO_TAILSPLIT") != nullptr;
    const ConvArgs& a = g.p[0];
    const int TM = cnk::conv_cfg_tm(cfg), TN = cnk::conv_cfg_tn(cfg), KS = cnk::conv_cfg_ks(cfg);
    const long long tiles = (long long)((a.n * a.T + TM - 1) / TM) * ((a.Cout + TN - 1) / TN);
    const int nks = a.ktaps * ((a.Cin_pad + KS - 1) / KS);
    const long long rem = tiles % cus_all;
    // (a fixed-plan stream-set has no split tail: WHICH tiles would be split follows a slot's position in the active list)
    if (!off && !fixed_plan && tiles > cus_all && rem > 0 && tiles <= sk_max_tiles) {
      int S = (int)std::min<long long>(8, cus_all / rem);
      S = std::min(S, nks / 8);                                  // at least 8 K-steps per slice (ups[3], 8 K-steps in all: the hand-off costs more than the tail it removes, 27 against 23 us)
      while (S > 1 && rem * S * TM * TN > sk_slab_floats) --S;
      if (S >= 2) { g.ksplit = S; g.split_from = (int)(tiles - rem); }
    }
    // (Tall tiles for small-M x huge-K layers - ups.0 at 64 streams as 64- or 128-row tiles with EVERY tile cut into K slices, so that
    // the 67 MB of weights cross the L2s 4x or 2x instead of 8x - measured round 6: 91-92 us against 90-91 of the 32-row plan; the launch
    // is bound by the f32 MFMA, not by the weight traffic.)
  }
  if (cnk::conv_cfg_tm(cfg) == 32) {
    // (grouped launches: one split factor for all problems, sized by the longest K loop)
    const int TM = cnk::conv_cfg_tm(cfg), TN = cnk::conv_cfg_tn(cfg);
    const int KS = cnk::conv_cfg_ks(cfg);
    long long tiles = 0;
    int nks = 0;
    for (int p = 0; p < nprob; ++p) {
      const ConvArgs& a = g.p[p];
      tiles += (long long)((plan_n(a.n) * a.T + TM - 1) / TM) * ((a.Cout + TN - 1) / TN);      // (one factor for every tile: fixed-plan stream-sets size it by max_slots)
      nks = std::max(nks, a.ktaps * ((a.Cin_pad + KS - 1) / KS));
    }
    int S = tiles > 0 ? (int)(ctx->num_cu / tiles) : 1;
    if (S > nks / 2) S = nks / 2;
    if (S > 16) S = 16;
    while (S > 1 && tiles * S * TM * TN > sk_slab_floats) --S;
    // the hand-off (write-through partial-tile stores, ticket, sc1 loads) is not free - with fences it cost about 6 K-steps
    // of 32 channels -: split only when it removes clearly more than that from the critical path (thresholds of 2 .. 12
    // measure the same at one and four streams)
    static const int min_saved = ch::dev_getenv("CONAN_SK_MIN") ? atoi(ch::dev_getenv("CONAN_SK_MIN")) : 12;
    if (S >= 2 && tiles <= sk_max_tiles && (nks - nks / S) * (KS / 32) >= min_saved) g.ksplit = S;
  }
  double fl = 0.0;
  for (int p = 0; p < nprob; ++p) fl += 2.0 * (double)g.p[p].n * g.p[p].T * g.p[p].Cout * g.p[p].ktaps * g.p[p].Cin;
  // (pipelined vocoder: its persistent launches leave `reserve_cus` CUs to the other internal streams, see launch_rb)
  profiled(cnk::conv_cfg_name(cfg), fl, st, [&] { cnk::launch_conv(g, nprob, cfg, st, cus_all); });
}

cnk::RowConvArgs conan_streams::mk_rc(const PackedConv& pc, const TRef& x, const TRef& y, int n, int T, int dil) const {
  cnk::RowConvArgs a; memset(&a, 0, sizeof(a));
  a.x = x; a.hist = ch::null_ref(); a.y = y; a.res = ch::null_ref(); a.m1 = ch::null_ref(); a.m2 = ch::null_ref();
  a.lnmask = ch::null_ref(); a.mask_out = ch::null_ref();
  a.w = pc.wf; a.bias = pc.bias; a.slots = d_slots; a.pos = pos_dec;
  a.Cin = pc.Cin; a.Cout = pc.Cout; a.Cout_pad = pc.wf_cout_pad; a.ktaps = pc.k; a.dil = dil; a.T = T; a.n = n;
  a.out_act = cnk::ACT_NONE; a.out_scale = 1.f; a.out_slope = 0.f; a.eps = 1e-5f;
  return a;
}

void conan_streams::rowconv(const cnk::RowConvArgs& a, hipStream_t st) {
  const double fl = 2.0 * (double)a.n * a.T * a.Cout * a.ktaps * a.Cin;
  if (mega_rec) {      // recording the decoder step's operator list (decoder_mega.hip)
    cnk::MegaOp op; memset(&op, 0, sizeof(op));
    op.u.rc = a;
    int ldsf = 0;
    const int prows = fixed_plan ? plan_n(a.n) * a.T : 0;
    const int v = cnk::rowconv_plan(op.u.rc, &op.nbx, &op.nby, &ldsf, prows);
    op.type = v == 3 ? cnk::MOP_ROWLIN : (v == 2 ? cnk::MOP_RC114 : cnk::MOP_RC111);
    // Several row tiles, a narrow layer (at most half as many 64-column strips as the group has members: the uv predictor's 128-wide
    // convs, the 256-wide 1x1 / k3 layers): 16-column strips with the K groups split over a workgroup's waves instead - every member
    // works, a wave's chain of MFMAs is a quarter as long - where the K-split patch (3 KB) still fits under the launch's LDS contract
    // (the k = 5, 256-channel window: tests/test_kernel_resources.py).  Results differ from the 64-column form by fp32 re-association.
    // (measured per operator at 64 streams, us: the uv predictor's 128 -> 128 k5 convs 9.0 -> 5.6, mel_out 6.4 -> 4.7, the 256 -> 256 k3 convs
    // 10.5 / 12.0 -> 9.8 / 10.6 - but the 256 -> 256 1x1 layers 6.5 -> 8.0: four waves' partial tiles through LDS for 16 K groups each
    // is more than the chain it shortens.  So: a quarter as many strips as members, or half as many with at least 3 taps.)
    const int strips64 = (op.u.rc.Cout_pad + 63) / 64;
    if (v == 0 && !a.w2 && plan_n(a.n) * a.T > 16 && mega_narrow_ksplit && (strips64 * 4 <= mega_gs || (strips64 * 2 <= mega_gs && a.ktaps >= 3)) && (a.ktaps * (a.Cin >> 4)) % 4 == 0) {
      const int ldsk = (32 + op.u.rc.wr_max * (a.Cin + 8) + 3 * 64 * 4);      // rowconv_lds_bytes(a, true) / 4
      cnk::MegaOp t = op; t.type = cnk::MOP_RC114;
      if (cnk::decoder_mega_lds_floats(t, ldsk) * 4 <= (96 + 32 * 264) * 4) {
        op.type = cnk::MOP_RC114; op.nbx = (op.u.rc.Cout_pad + 15) / 16; ldsf = ldsk;
      }
    }
    if (a.w2) {        // fused conv -> 1x1 conv: behind the window, the member's 16 x (Cout / 8 + 8) hidden tile - or, with a single 64-column
                       // hidden strip per member (the decoder's conv blocks), in the window's place once every wave has read it out
      // (v == 2: a single row tile - the plan of the plain conv would split K over the waves; the fused operator has its own geometry)
      if ((v != 0 && v != 2) || (a.ktaps * (a.Cin >> 4)) % 8 != 0) { mega_rec_ok = false; return; }
      op.type = cnk::MOP_FFN;
      const int hc = a.Cout / mega_ffn_gs;
      op.u.rc.hid_overlay = hc == 64 ? 1 : 0;
      ldsf = op.u.rc.hid_overlay ? std::max(op.u.rc.wr_max * (a.Cin + 8), 16 * (hc + 8)) : op.u.rc.wr_max * (a.Cin + 8) + 16 * (hc + 8);
      mega_rec_flops += 2.0 * (double)a.n * a.T * a.Cout * a.Cout2;
    }
    mega_rec_flops += fl;
    // (the megakernel places its row table in front of the operator's window: the stand-alone kernel's size + ROWTAB_FLOATS)
    mega_push(op, cnk::decoder_mega_lds_floats(op, ldsf));
    return;
  }
  const int prows = fixed_plan ? plan_n(a.n) * a.T : 0;
  profiled(cnk::rowconv_kernel_name(a, prows), fl, st, [&] { cnk::launch_rowconv(a, st, prows); });
}

// one ResBlock1 unit per branch (c1 -> LeakyReLU -> c2 -> + residual) as one tile pass
bool conan_streams::launch_rb(const cnk::RBArgs& ain, int C, hipStream_t st, const TRef* ymean) {
  cnk::RBArgs a = ain;
  a.sched = rb_sched[ws_index(st) == 1 ? 1 : 0];
  int ksum = 0, kmax = 0;
  double fl = 0.0;
  for (int p = 0; p < a.nprob; ++p) { ksum += a.p[p].k; kmax = std::max(kmax, a.p[p].k); fl += 2.0 * 2.0 * (double)a.n * a.T * C * C * a.p[p].k; }
  // CUs left to the other internal stream while a pipelined step is in flight (conan_step_async): the persistent
  // blocks of this launch hold their CU for its whole duration, so without a few free CUs every one of the ~60
  // dependent front-end launches of the next chunk waits for a vocoder kernel boundary
  const int cus = std::max(8, ctx->num_cu - (ws_index(st) == 1 ? reserve_cus : 0));
  // bf16-limb form (resblock_limb.hip) where the weights were packed for it and every branch's span fits its window
  int span = 0;
  bool limb = rb_limb;
  for (int p = 0; p < a.nprob; ++p) { span = std::max(span, (a.p[p].k - 1) * a.p[p].dil); limb = limb && a.p[p].w1l && a.p[p].w2l; }
  limb = limb && plan_n(a.n) <= cnk::kResblockLimbMaxSlots && cnk::resblock_limb_supported(C, kmax, span);
  // (AUTO takes the limb pass wherever it exists, like an explicit request: measured at 1 / 4 / 8 / 16 / 32 streams the limb pass is
  // never the slower one - 0.72 against 0.87 ms per pipelined step at 8 streams, equal at one - tools/arith_sweep.sh)
  const int rows = limb ? cnk::resblock_limb_rows(C, span) : cnk::resblock_fused_rows(C, a.T, plan_n(a.n), ksum, kmax, cus);
  // Last dilation of a stage: with at least one (slot, row tile) group per CU a workgroup runs the group's branches one after
  // the other and stores only leaky_relu(mean) - equal work per group, no branch outputs written, no mean_act launch.
  // With fewer groups than CUs (C = 128 at 64 streams: 128) the branches stay separate tiles.
  const long long groups = (long long)plan_n(a.n) * ((a.T + rows - 1) / rows);
  // (a group is the unit of work of a merged launch: the last round of groups must not leave most CUs idle - 320 groups on 256
  // CUs, the limb build of the C = 128 stage at 64 streams, ran 137 us merged against 98 + 6 us as separate branches + mean_act)
  const long long rounds = (groups + cus - 1) / cus;
  const bool balanced = groups * 10 >= rounds * cus * 9;
  a.merge = (ymean && rb_merge && a.nprob > 1 && groups >= cus && balanced && (limb ? cnk::resblock_limb_can_merge(C, rows) : cnk::resblock_fused_can_merge(C, rows))) ? 1 : 0;
  if (a.merge) a.ymean = *ymean;
  profiled(limb ? cnk::resblock_limb_name(C, rows, a.merge != 0) : cnk::resblock_fused_name(C, rows, a.merge != 0), fl, st, [&] {
    if (!(limb ? cnk::launch_resblock_limb(a, C, rows, cus, st) : cnk::launch_resblock_fused(a, C, rows, cus, st))) throw Error(CONAN_ERR_HIP, "fused resblock launch failed");
  });
  return a.merge != 0;
}

// one ResBlock1 unit per branch of the wide first stage, a pair of workgroups per (branch, stream) tile
void conan_streams::launch_rp(const cnk::RPArgs& ain, hipStream_t st) {
  cnk::RPArgs a = ain;
  const int w = ws_index(st) == 1 ? 1 : 0;
  const int pairs = ctx->num_cu / 2;
  a.xb = rp_xb[w]; a.xflag = rp_words[w]; a.mbox = rp_words[w] + (size_t)pairs * 8; a.xcount = rp_words[w] + (size_t)pairs * 12;
  a.sched = reinterpret_cast<int*>(rp_words[w] + (size_t)pairs * 14);
  a.guard = d_guard; a.fault = test_fault == 3 ? 1 : 0;
  if (test_fault == 3) test_fault = 0;
  double fl = 0.0;
  for (int p = 0; p < a.nprob; ++p) fl += 2.0 * 2.0 * (double)a.n * a.T * 256.0 * 256.0 * a.p[p].k;
  profiled(cnk::resblock_pair_name(a.T), fl, st, [&] {
    if (!cnk::launch_resblock_pair(a, ctx->num_cu, st)) throw Error(CONAN_ERR_HIP, "resblock pair launch failed");
  });
}

void conan_streams::mega_print_stamps() {
  if (!mega_dbg || mega_dbg_prog < 0 || mega_dbg_prog >= (int)mega_cache.size()) return;
  (void)hipDeviceSynchronize();
  const MegaProgram& e = mega_cache[mega_dbg_prog];
  std::vector<unsigned long long> h(cnk::kMegaDbgWords);
  if (hipMemcpy(h.data(), mega_dbg, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
  static const char* names[] = {"rowconv<1,1,1>", "rowconv<1,1,4>", "rowlin", "layernorm", "xattn", "pitch_head", "embed", "copy32", "advance", "ffn"};
  constexpr int nnames = (int)(sizeof(names) / sizeof(names[0]));
  fprintf(stderr, "[decoder_mega] last launch: %d groups x %d workgroups, %d jobs, %d operators, %d group barriers, %.1f us in all (workgroup 0; first job per operator below)\n",
          e.groups, e.group_size, e.njobs, e.nops, e.barriers, (h[e.nops + 1] - h[0]) / 100.0);
  if (!e.xcd) {
    fprintf(stderr, "  groups on one XCD (l2 mode) / XCC masks:");
    for (int g2 = 0; g2 < e.groups && g2 < 32; ++g2) if (h[cnk::kMegaDbgGroupWords + g2]) fprintf(stderr, " %d:%llu/%02llx", g2, h[cnk::kMegaDbgGroupWords + g2] & 1ull, (h[cnk::kMegaDbgGroupWords + g2] >> 16) & 0xffull);
    fprintf(stderr, "\n");
  }
  for (int o = 0; o < e.nops; ++o) {
    const cnk::MegaOp& op = e.pinned[o];
    fprintf(stderr, "  op %2d %-15s strips %4d  barrier %d  %7.2f us", o, (op.type >= 0 && op.type < nnames) ? names[op.type] : "?", op.nbx, op.barrier, (h[o + 1] - h[o]) / 100.0);
    if (op.type <= cnk::MOP_ROWLIN) fprintf(stderr, "   Cin %4d Cout %4d k %d ln %d", op.u.rc.Cin, op.u.rc.Cout, op.u.rc.ktaps, op.u.rc.ln);
    if ((op.type == cnk::MOP_RC114 || op.type == cnk::MOP_RC111) && h[128 + o * 4]) fprintf(stderr, "   [args+warm %.2f stage %.2f strips %.2f barrier %.2f]", (h[128 + o * 4] - h[o]) / 100.0, (h[128 + o * 4 + 1] - h[128 + o * 4]) / 100.0,
                                                               (h[128 + o * 4 + 2] - h[128 + o * 4 + 1]) / 100.0, (h[o + 1] - h[128 + o * 4 + 2]) / 100.0);
    if (op.type == cnk::MOP_RC114 && h[512 + o * 4]) fprintf(stderr, " {stage: sync %.2f issue %.2f return+lds %.2f sync %.2f rest %.2f}", (h[512 + o * 4] - h[128 + o * 4]) / 100.0, (h[512 + o * 4 + 1] - h[512 + o * 4]) / 100.0,
                                                               (h[512 + o * 4 + 2] - h[512 + o * 4 + 1]) / 100.0, (h[512 + o * 4 + 3] - h[512 + o * 4 + 2]) / 100.0, (h[128 + o * 4 + 1] - h[512 + o * 4 + 3]) / 100.0);
    fprintf(stderr, "\n");
  }
}

void conan_streams::launch_mega(MegaProgram& e, hipStream_t st) {
  static const bool stamps = ch::dev_getenv("CONAN_MEGA_STAMPS") != nullptr;
  if (stamps && !mega_dbg) mega_dbg = reinterpret_cast<unsigned long long*>(alloc(2 * (size_t)cnk::kMegaDbgWords));
  if (stamps) HIP_CHECK(hipMemsetAsync(mega_dbg, 0, sizeof(unsigned long long) * cnk::kMegaDbgWords, st));      // (no stale stamps of another program)
  if (stamps) mega_dbg_prog = (int)(&e - mega_cache.data());
  cnk::MegaLaunch m; memset(&m, 0, sizeof(m));
  m.prog = e.dev; m.nops = e.nops; m.njobs = e.njobs; m.groups = e.groups; m.group_size = e.group_size; m.kw4 = e.kw4; m.lds_bytes = e.lds_bytes;
  m.slots = d_slots; m.pos = pos_dec; m.n = e.n; m.T = e.T;
  m.gbar = mega_bar + 16; m.bar = mega_bar; m.bar_base = mega_bar_count; m.dbg = mega_dbg; m.guard = d_guard; m.wide_regs = rb_limb ? 1 : 0;
  m.xcd = e.xcd ? 1 : 0;
  const bool pipelined = st_front != nullptr && st == st_front;
  if (e.xcd) {
    // Pipelined single-tile steps: the group's members take only the LDS their operators need (36.5 KB), so that the small-batch
    // vocoder's conv_mfma workgroups (112.5 KB since round 6) find room beside them (round 5 launched 84 KB here as well: 32 blocks of
    // every vocoder launch then waited for the decoder step to end - 0.635 ms per one-stream pipelined step, 0.40 now).  Same program,
    // same bits.
    static const bool pad_always = ch::dev_getenv("CONAN_MEGA_XCD_PAD") != nullptr;
    if (pipelined && !pad_always) m.lds_bytes = e.lds_need;
    mega_xseq = mega_xseq + 1u;       // (20 bits of it travel in the election word; consecutive launches differ)
    m.xs = mega_x; m.xseq = mega_xseq; m.xdec_base = mega_xdec;
    if (test_fault == 1) { m.xseq += 7u; mega_xseq += 7u; test_fault = 0; }     // test hook: the election word is not in the state this launch expects - nobody can claim
    profiled("cnk::decoder_mega_kernel<4, 2>", e.flops, st, [&] { cnk::launch_decoder_mega(m, st); });
    mega_xdec += (unsigned)(e.groups * e.group_size);
    return;
  }
  if (test_fault == 1) { m.bar_base += 1u; test_fault = 0; }       // test hook: the grid barrier waits for one arrival too many
  mega_gseq = mega_gseq + 1u;             // (its own sequence: the xcd mode's election word tracks mega_xseq launch by launch)
  m.xs = mega_x; m.xseq = mega_gseq;      // (the flag barriers of groups that sit on one XCD count in epochs of this sequence number)
  { const bool nol2 = dev("MEGA_NOL2") != nullptr;
    // Layout of the multi-tile launch's groups: group-fastest (a group on one XCD, hand-offs through its L2) for every step.
    // Member-fastest (member s of every group on XCD s: strip s's weights stay in that XCD's L2 for all groups - 159 instead of
    // 387 MB fetched per step, profiles/r6_*) for PIPELINED steps was tried in round 6 as VERDICT round 5 asked: the step time is the
    // same (1.3958 against 1.3984 ms, three alternating runs), the bits are the same - and with two stream-sets overlapping on the
    // device (tests/test_gpu_stress.py, 128 streams, f32) the pair kernel's bounded partner wait gave up in 3 of 8 runs and one run
    // ended in a GPU memory fault, against 0 of 16 with this layout (tools/stress_repro.py; profiles/r6_stress_layout.txt).  The
    // bytes are not worth a launch shape whose forward progress beside other waiting launches is not understood: MEGA_LAYOUT=m
    // stays a developer switch.
    const char* lay = dev("MEGA_LAYOUT");
    const bool mfast = lay && lay[0] == 'm';
    m.xdec_base = (nol2 ? 1u : 0u) | (mfast ? 2u : 0u); }
  profiled(rb_limb ? "cnk::decoder_mega_kernel<4, 3>" : "cnk::decoder_mega_kernel<6, 3>", e.flops, st, [&] { cnk::launch_decoder_mega(m, st); });
  mega_bar_count += (unsigned)(e.groups * e.group_size);
}

// conan_streams_opts.dev_plan: "NAME=value;NAME=value" (a bare NAME means NAME=1)
void conan_streams::parse_dev_plan(const char* text) {
  static const char* known[] = {"RESERVE_CUS", "ROWCONV", "RB_NOMERGE", "RB_NOLIMB", "FENCED", "DEC_MEGA", "MEGA_GRID", "FRONT_CUSTRIDE", "EMF_CUSTRIDE", "MEGA_GS",
                                "MEGA_NARROW", "MEGA_NOL2", "MEGA_LAYOUT", "FRONT_PRIO", "RB_UNFUSED", "RB_FUSED", "RB_PAIR", "RB_NOPAIR", "RP_MIN_SLOTS",
                                "UPS_CFG", "EMF_CLUSTER", "EMF_UNFUSED", "NO_TALL", "TALL_MAXT"};
  dev_plan.clear();
  if (!text) return;
  const std::string t(text);
  size_t p = 0;
  while (p < t.size()) {
    size_t q = t.find(';', p);
    if (q == std::string::npos) q = t.size();
    std::string item = t.substr(p, q - p);
    p = q + 1;
    while (!item.empty() && item.front() == ' ') item.erase(item.begin());
    while (!item.empty() && item.back() == ' ') item.pop_back();
    if (item.empty()) continue;
    const size_t eq = item.find('=');
    const std::string name = item.substr(0, eq), value = eq == std::string::npos ? "1" : item.substr(eq + 1);
    bool ok = false;
    for (const char* k : known) ok = ok || name == k;
    if (!ok) throw Error(CONAN_ERR_INVALID, "conan_streams_opts.dev_plan: unknown switch '" + name + "'");
    dev_plan[name] = value;
  }
}

void conan_streams::set_slots(const int32_t* slots, int n, hipStream_t st) {
  if (n <= 0 || n > max_slots) throw Error(CONAN_ERR_INVALID, "slot count out of range");
  bool same = (int)h_slots.size() == n;
  for (int i = 0; i < n; ++i) {
    if (slots[i] < 0 || slots[i] >= max_slots) throw Error(CONAN_ERR_INVALID, "slot index out of range");
    if (same && h_slots[i] != slots[i]) same = false;
  }
  if (same) return;
  if (++slot_gen == 0x7fffffff) { std::fill(slot_seen.begin(), slot_seen.end(), 0); slot_gen = 1; }
  for (int i = 0; i < n; ++i) {
    if (slot_seen[slots[i]] == slot_gen) throw Error(CONAN_ERR_INVALID, "duplicate slot");
    slot_seen[slots[i]] = slot_gen;
  }
  h_slots.assign(slots, slots + n);
  // (entries n .. n + kSlotTablePad - 1 = copies of the last slot: a ragged last tile of conv_limb stages ALL the slots it has room for -
  // TM / T of them, e.g. 4-5 with 40 ms chunks in the C = 256 stage - and reads the table up to TM / T - 1 entries past n)
  std::vector<int> up(h_slots); up.resize((size_t)n + cnk::kSlotTablePad, h_slots.back());
  pin.upload(d_slots, up.data(), up.size(), st);
}

// ------------------------------------------------------------------------------------------------ pipelined stepping

void conan_streams::async_init() {
  if (st_front) return;
  {
    // CONAN_FRONT_PRIO: 1 = front-end stream at the highest priority, -1 = at the lowest, 0 = default (experiment knob)
    int lo = 0, hi = 0;
    HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    const char* e = dev("FRONT_PRIO");
    const int mode = e ? atoi(e) : 0;
    // CONAN_FRONT_CUSTRIDE / CONAN_EMF_CUSTRIDE = s: that stage's stream may only use CUs i with i % s == 0 (experiment knob)
    auto masked = [&](hipStream_t* st, const char* env) {
      const char* v = dev(env);
      const int stride = v ? atoi(v) : 0;
      if (stride < 2) { HIP_CHECK(hipStreamCreateWithFlags(st, hipStreamNonBlocking)); return; }
      std::vector<uint32_t> mask((ctx->num_cu + 31) / 32, 0u);
      for (int i = 0; i < ctx->num_cu; i += stride) mask[i / 32] |= 1u << (i % 32);
      HIP_CHECK(hipExtStreamCreateWithCUMask(st, (uint32_t)mask.size(), mask.data()));
    };
    if (mode == 0) {
      masked(&st_front, "FRONT_CUSTRIDE");
      HIP_CHECK(hipStreamCreateWithFlags(&st_voc, hipStreamNonBlocking));
      masked(&st_emf, "EMF_CUSTRIDE");
    } else {
      HIP_CHECK(hipStreamCreateWithPriority(&st_emf, hipStreamNonBlocking, mode > 0 ? hi : lo));
      HIP_CHECK(hipStreamCreateWithPriority(&st_front, hipStreamNonBlocking, mode > 0 ? hi : lo));
      HIP_CHECK(hipStreamCreateWithPriority(&st_voc, hipStreamNonBlocking, mode > 0 ? lo : hi));
    }
  }
  for (int i = 0; i < NP; ++i) {
    HIP_CHECK(hipEventCreateWithFlags(&ev_in[i], hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&ev_fence[i], hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&ev_emf[i], hipEventDisableTiming));
    codes_hand[i] = (int*)alloc((size_t)max_slots * max_frames);
    HIP_CHECK(hipEventCreateWithFlags(&ev_front[i], hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&ev_voc[i], hipEventDisableTiming));
    mel_hand[i] = alloc((size_t)max_slots * max_frames * ctx->cfg.num_mels);
  }
}

void conan_streams::join(hipStream_t st) {
  if (async_steps == 0) return;
  const int last = (int)((async_steps - 1) % NP);       // the internal streams are in-order: the last step covers all
  HIP_CHECK(hipStreamWaitEvent(st, ev_emf[last], 0));
  HIP_CHECK(hipStreamWaitEvent(st, ev_front[last], 0));
  HIP_CHECK(hipStreamWaitEvent(st, ev_voc[last], 0));
}

// ------------------------------------------------------------------------------------------------ vocoder

void conan_streams::build_vocoder() {
  const conan_cfg& c = ctx->cfg;
  int maxk = 0;
  for (int b = 0; b < c.voc_num_resblocks; ++b) maxk = std::max(maxk, c.voc_rb_kernels[b]);
  v_mel = mk_ring(c.num_mels, 1, 6, &voc_state);
  v_pre = mk_ring(c.voc_initial_channel, 1, c.voc_up_kernels[0] - 1, &voc_state);
  int ch_ = c.voc_initial_channel, rate = 1;
  v_st.resize(c.voc_num_ups);
  for (int i = 0; i < c.voc_num_ups; ++i) {
    VocStage& s = v_st[i];
    rate *= c.voc_up_rates[i];
    ch_ /= 2;
    s.C = ch_; s.rate = rate;
    // ResBlock1 stages whose width the fused tile pass covers: c1's halo rows of xt are recomputed from the input ring,
    // so an input keeps (k-1)*(dil+1) rows of history and neither xt nor activated twins exist
    // (stream-sets of a few slots cannot fill the chip with whole-width tiles: they keep the two-launch plan for the wide
    // stages; at C = 32 a tile is short enough that the single pass wins even for one stream: 0.99 -> 0.95 ms per chunk)
    // (limb stream-sets: from 4 slots - measured per chunk at 4 / 6 streams 0.591 / 0.676 ms fused against 0.603 / 0.714 ms with the
    // two-launch plan in the wide stages, blocking p50 0.98 / 1.14 against 1.03 / 1.19 ms; at 1-3 streams the two-launch plan is as fast or faster)
    s.fused = c.voc_resblock != 2 && dev("RB_UNFUSED") == nullptr && (max_slots >= (rb_limb ? 4 : 8) || ch_ <= 32 || dev("RB_FUSED") != nullptr);
    for (int b = 0; b < c.voc_num_resblocks && s.fused; ++b)
      for (int d = 0; d < c.voc_rb_num_dil; ++d)
        s.fused = s.fused && cnk::resblock_fused_supported(ch_, c.voc_rb_kernels[b], (c.voc_rb_kernels[b] - 1) * c.voc_rb_dilations[b][d]);
    // The wide first stage (C = 256): too few rows per stream for whole-width tiles - pairs of workgroups per (branch, stream)
    // tile (resblock_pair.hip).  A tile is all rows of a stream's step, so the stream-set must not take more than 32 rows per
    // step in this stage (windowed / whole-utterance stream-sets keep the two-launch plan), and it needs enough streams to
    // give the chip tiles.
    // ... for exact-f32 stream-sets.  bf16-limb stream-sets of the same sizes run this stage's convs as conv_limb's grouped launches
    // (three problems of 3 / 7 / 11 taps per launch, 64-row tiles: 6 tiles per stream): 6 x 33-44 us against the pair kernel's 3 x 96-99,
    // which is the time of ONE tile at any stream count below 128 - measured per pipelined step / blocking p50 at 16 / 24 / 32 / 40
    // streams 0.806 -> 0.760 / 1.45 -> 1.36, 0.906 -> 0.818 / 1.58 -> 1.48, 0.992 -> 0.949 / 1.66 -> 1.58, 1.170 -> 1.083 / 1.89 -> 1.79 ms
    // (round 4, late; until then from 48 slots on, where the two took the same time alone before conv_limb's last 10 %).  Below 16
    // slots the stage keeps conv_mfma's two-launch plan with its split-K tails: the grouped limb launches would be 2-5 % faster per
    // pipelined step there and 5-9 % slower per blocking step.  CONAN_RB_PAIR=1 keeps the pair kernel.
    const bool limb_groups = rb_limb && max_slots >= 16 && dev("RB_PAIR") == nullptr;
    if (!s.fused && !limb_groups && c.voc_resblock != 2 && dev("RB_UNFUSED") == nullptr && dev("RB_NOPAIR") == nullptr &&
        c.voc_num_resblocks <= kMaxBranches && max_slots >= (dev("RP_MIN_SLOTS") ? atoi(dev("RP_MIN_SLOTS")) : 16) && max_frames * rate <= 32) {
      s.pair = true;
      for (int b = 0; b < c.voc_num_resblocks; ++b)
        for (int d = 0; d < c.voc_rb_num_dil; ++d)
          s.pair = s.pair && cnk::resblock_pair_supported(ch_, c.voc_rb_kernels[b], (c.voc_rb_kernels[b] - 1) * c.voc_rb_dilations[b][d], max_frames * rate);
      s.fused = s.pair;      // ring plan of the fused stages: raw tensors only
    }
    int up_hist = (maxk - 1) * c.voc_rb_dilations[0][0];
    if (s.fused) for (int b = 0; b < c.voc_num_resblocks; ++b) up_hist = std::max(up_hist, (c.voc_rb_kernels[b] - 1) * (c.voc_rb_dilations[b][0] + 1));
    s.up = mk_ring(ch_, rate, up_hist, &voc_state);
    const int next_pad = (i + 1 < c.voc_num_ups) ? c.voc_up_kernels[i + 1] - 1 : 6;
    s.xs = mk_ring(ch_, rate, next_pad, &voc_state);
    // LeakyReLU'd twins of `up` and of the resblock outputs that feed another c1: the producer's epilogue writes both
    // (ConvArgs::y2_base), c1 reads the activated copy, the residual add reads the raw one
    if (!s.fused) s.upa = mk_ring(ch_, rate, (maxk - 1) * c.voc_rb_dilations[0][0], &voc_state);
    s.xt.resize(c.voc_num_resblocks); s.xo.resize(c.voc_num_resblocks); s.xa.resize(c.voc_num_resblocks);
    for (int b = 0; b < c.voc_num_resblocks; ++b)
      for (int d = 0; d < c.voc_rb_num_dil; ++d) {
        const int k = c.voc_rb_kernels[b];
        if (!s.fused) s.xt[b].push_back(mk_ring(ch_, rate, k - 1, &voc_state));
        const int h = (d + 1 < c.voc_rb_num_dil) ? (k - 1) * (c.voc_rb_dilations[b][d + 1] + (s.fused ? 1 : 0)) : next_pad;
        s.xo[b].push_back(mk_ring(ch_, rate, h, &voc_state));
        if (!s.fused && d + 1 < c.voc_rb_num_dil) s.xa[b].push_back(mk_ring(ch_, rate, h, &voc_state));
      }
    if (s.pair) {
      s.xh.resize(c.voc_num_resblocks);
      for (int b = 0; b < c.voc_num_resblocks; ++b)
        for (int d = 0; d < c.voc_rb_num_dil; ++d) s.xh[b].push_back(mk_ring(ch_, rate, c.voc_rb_kernels[b] - 1, &voc_state));
      for (int w = 0; w < 2 && !rp_xb[w]; ++w) {
        const int pairs = ctx->num_cu / 2;
        rp_xb[w] = alloc(cnk::resblock_pair_xb_floats(ctx->num_cu));
        rp_words[w] = reinterpret_cast<unsigned*>(alloc((size_t)pairs * 14 + 4));       // (alloc zero-fills)
      }
    }
  }
}

void conan_streams::hifigan_step(int n, int frames, const float* mel_dev, float* wav_out, float* pre_tanh, hipStream_t st, const conan_hifigan_taps* taps) {
  const conan_cfg& c = ctx->cfg;
  const int* pos = pos_voc;
  const float LR = 0.1f;   // LRELU_SLOPE, hifigan_causal.py:20
  // upsample 'nn' (CausalUpsampleBlock1) reads input frames AHEAD of the output frame and zeros beyond the last one:
  // a step is a whole forward (utterance or window) from freshly reset state, whose untouched ring rows are those zeros
  const bool lookahead = c.voc_upsample == 2;
  for (int i = 0; i < n; ++i) {
    if (lookahead && !voc_fresh[h_slots[i]])
      throw Error(CONAN_ERR_STATE, "upsample 'nn' (CausalUpsampleBlock1) looks ahead of the frame it writes: slot " + std::to_string(h_slots[i]) +
                                       " needs conan_streams_reset(CONAN_MODEL_HIFIGAN) before every vocoder step (whole-utterance or windowed forward only)");
    voc_fresh[h_slots[i]] = 0;
  }
  {  // mel chunk -> ring (conv_pre needs 6 frames of left context)
    cnk::CopyArgs ca; memset(&ca, 0, sizeof(ca));
    ca.x = ch::lin_ref(const_cast<float*>(mel_dev), frames, c.num_mels); ca.y = v_mel.ref();
    ca.slots = d_slots; ca.pos = pos; ca.lens = nullptr; ca.T = frames; ca.n = n; ca.C = c.num_mels;
    cnk::launch_copy_rows(ca, st);
  }
  {  // conv_pre; its only consumer is leaky_relu -> ups[0] (hifigan_causal.py:319-322), so the activation is stored
    ConvArgs a = mk(ctx->conv("voc.conv_pre"), v_mel.ref(), v_pre.ref(), n, frames, pos);
    a.out_act = cnk::ACT_LRELU; a.out_slope = LR;
    conv(a, st);
  }
  auto tap = [&](float* dst, const Ring& r, int T) {   // rows of this step: ring -> caller buffer [n][T][C]
    if (!dst) return;
    cnk::CopyArgs ca; memset(&ca, 0, sizeof(ca));
    ca.x = r.ref(); ca.y = ch::lin_ref(dst, T, r.C); ca.slots = d_slots; ca.pos = pos; ca.T = T; ca.n = n; ca.C = r.C;
    cnk::launch_copy_rows(ca, st);
  };
  if (taps) tap(taps->conv_pre_act, v_pre, frames);
  const int NB = c.voc_num_resblocks, ND = c.voc_rb_num_dil;
  if (NB > kMaxBranches) throw Error(CONAN_ERR_UNSUPPORTED, "more than 3 resblock branches");
  int ridx = 0;
  bool last_merged = false;
  for (int i = 0; i < c.voc_num_ups; ++i) {
    VocStage& s = v_st[i];
    bool merged = false;                    // the stage's branch mean was formed by its last fused launch
    const int Tin = frames * (s.rate / c.voc_up_rates[i]);
    const int T = frames * s.rate;
    {  // x = leaky_relu(x); x = ups[i](x)   (hifigan_causal.py:321-322): the input (v_pre / branch mean xs) is stored
       // activated; the output goes out raw (residual operand) and activated (c1 operand)
      ConvArgs a = mk(ctx->conv("voc.ups." + std::to_string(i)), i == 0 ? v_pre.ref() : v_st[i - 1].xs.ref(), s.up.ref(), n, Tin, pos, 1,
                      lookahead ? 0 : -1);
      if (!s.fused) { a.y2_base = s.upa.base; a.y2_slope = LR; }
      {
        // developer switch: CONAN_UPS_CFG=c0,c1,c2,c3 forces the tile configuration (ConvCfg index) of the i-th upsampler
        const std::vector<int> forced = [&] { std::vector<int> v; const char* e = dev("UPS_CFG"); if (e) { std::string t(e); size_t p = 0; while (p <= t.size()) { size_t q = t.find(',', p); if (q == std::string::npos) q = t.size(); v.push_back(atoi(t.substr(p, q - p).c_str())); p = q + 1; } } return v; }();
        if (i < (int)forced.size() && forced[i] >= 0 && forced[i] < cnk::NUM_CFG) { ConvGroup g; g.p[0] = a; launch_group(g, 1, forced[i], st); }
        else conv(a, st);
      }
      if (taps) tap(taps->ups[i], s.up, T);
    }
    for (int d = 0; d < ND && s.pair; ++d) {    // ResBlock1 of the wide first stage: pairs of workgroups per (branch, stream) tile
      if (T > 32) throw Error(CONAN_ERR_INVALID, "more frames in a step than this stream-set was created for");
      cnk::RPArgs ra; memset(&ra, 0, sizeof(ra));
      for (int b = 0; b < NB; ++b) {
        const std::string base = "voc.rbf." + std::to_string(ridx + b);
        cnk::RPProb& pr = ra.p[b];
        pr.w1 = ctx->vec(base + ".c1." + std::to_string(d) + ".w"); pr.b1 = ctx->vec(base + ".c1." + std::to_string(d) + ".b");
        pr.w2 = ctx->vec(base + ".c2." + std::to_string(d) + ".w"); pr.b2 = ctx->vec(base + ".c2." + std::to_string(d) + ".b");
        pr.x = d == 0 ? s.up.ref() : s.xo[b][d - 1].ref();
        pr.y = s.xo[b][d].ref();
        pr.xh = s.xh[b][d].ref();
        pr.k = c.voc_rb_kernels[b]; pr.dil = c.voc_rb_dilations[b][d];
      }
      ra.slots = d_slots; ra.pos = pos; ra.nprob = NB; ra.n = n; ra.T = T; ra.slope = LR;
      launch_rp(ra, st);
      if (d + 1 == ND && mark_wide) { HIP_CHECK(hipEventRecord(mark_wide, st)); for (int q = 0; q < 4; ++q) if (ev_wide[q] == mark_wide) wide_marked[q] = true; }
    }
    for (int d = 0; d < ND && s.fused && !s.pair; ++d) {   // ResBlock1 (hifigan_causal.py:230-238): c1 -> lrelu -> c2 -> + x in one tile pass per branch
      cnk::RBArgs ra; memset(&ra, 0, sizeof(ra));
      for (int b = 0; b < NB; ++b) {
        const std::string base = "voc.rbf." + std::to_string(ridx + b);
        cnk::RBProb& pr = ra.p[b];
        pr.w1 = ctx->vec(base + ".c1." + std::to_string(d) + ".w"); pr.b1 = ctx->vec(base + ".c1." + std::to_string(d) + ".b");
        pr.w2 = ctx->vec(base + ".c2." + std::to_string(d) + ".w"); pr.b2 = ctx->vec(base + ".c2." + std::to_string(d) + ".b");
        pr.w1l = reinterpret_cast<const unsigned short*>(ctx->vec_or_null(base + ".c1." + std::to_string(d) + ".wl"));
        pr.w2l = reinterpret_cast<const unsigned short*>(ctx->vec_or_null(base + ".c2." + std::to_string(d) + ".wl"));
        pr.x = d == 0 ? s.up.ref() : s.xo[b][d - 1].ref();
        pr.y = s.xo[b][d].ref();
        pr.k = c.voc_rb_kernels[b]; pr.dil = c.voc_rb_dilations[b][d];
      }
      ra.slots = d_slots; ra.pos = pos; ra.nprob = NB; ra.n = n; ra.T = T; ra.slope = LR;
      if (d + 1 < ND) launch_rb(ra, s.C, st);
      else { const TRef ym = s.xs.ref(); merged = launch_rb(ra, s.C, st, &ym); }
    }
    for (int d = 0; d < ND && c.voc_resblock == 2; ++d) {  // ResBlock2 (hifigan_causal.py:255-261): x = conv_d(lrelu(x)) + x
      ConvGroup g1;
      for (int b = 0; b < NB; ++b) {
        const TRef xin = d == 0 ? s.up.ref() : s.xo[b][d - 1].ref();
        const TRef xin_act = d == 0 ? s.upa.ref() : s.xa[b][d - 1].ref();
        ConvArgs a1 = mk(ctx->conv("voc.rb." + std::to_string(ridx + b) + ".c." + std::to_string(d)), xin_act, s.xo[b][d].ref(), n, T, pos,
                         c.voc_rb_dilations[b][d]);
        a1.res = xin; a1.has_res = 1;
        if (d + 1 < ND) { a1.y2_base = s.xa[b][d].base; a1.y2_slope = LR; }
        g1.p[b] = a1;
      }
      launch_group(g1, NB, pick_cfg(plan_n(n) * T, s.C, NB), st);
    }
    for (int d = 0; d < ND && c.voc_resblock != 2 && !s.fused; ++d) {  // ResBlock1 as two grouped conv launches (widths the fused pass does not cover)
      ConvGroup g1, g2;
      for (int b = 0; b < NB; ++b) {
        // xt = c1(leaky_relu(x)); x = c2(leaky_relu(xt)) + x: both activations are applied where the tensor is written
        const TRef xin = d == 0 ? s.up.ref() : s.xo[b][d - 1].ref();
        const TRef xin_act = d == 0 ? s.upa.ref() : s.xa[b][d - 1].ref();
        std::string base = "voc.rb." + std::to_string(ridx + b);
        ConvArgs a1 = mk(ctx->conv(base + ".c1." + std::to_string(d)), xin_act, s.xt[b][d].ref(), n, T, pos, c.voc_rb_dilations[b][d]);
        a1.out_act = cnk::ACT_LRELU; a1.out_slope = LR;
        g1.p[b] = a1;
        ConvArgs a2 = mk(ctx->conv(base + ".c2." + std::to_string(d)), s.xt[b][d].ref(), s.xo[b][d].ref(), n, T, pos, 1);
        a2.res = xin; a2.has_res = 1;
        if (d + 1 < ND) { a2.y2_base = s.xa[b][d].base; a2.y2_slope = LR; }
        g2.p[b] = a2;
      }
      const int cfg = pick_cfg(plan_n(n) * T, s.C, NB);
      launch_group(g1, NB, cfg, st);
      launch_group(g2, NB, cfg, st);
    }
    // (a caller that taps the last stage's output gets it from the separate mean launch: conv_post, which otherwise forms the
    // mean itself, also advances the frame counters the tap's row addresses depend on)
    const bool tap_last = taps && taps->stage_out[i] && i + 1 == c.voc_num_ups && !merged;
    last_merged = merged || tap_last;
    if ((i + 1 < c.voc_num_ups || tap_last) && !merged) {  // xs = leaky_relu(mean_b ResBlock_b(x))   (hifigan_causal.py:324-331), consumed by ups[i+1]; the last stage's by conv_post, which forms it itself
      cnk::MeanActArgs ma; memset(&ma, 0, sizeof(ma));
      for (int b = 0; b < NB; ++b) ma.x[b] = s.xo[b][ND - 1].ref();
      ma.y = s.xs.ref(); ma.slots = d_slots; ma.pos = pos; ma.nsrc = NB; ma.T = T; ma.n = n; ma.C = s.C; ma.slope = LR;
      cnk::launch_mean_act(ma, st);
    }
    if (taps) tap(taps->stage_out[i], s.xs, T);
    ridx += NB;
  }
  {  // conv_post + tanh on leaky_relu(xs / NB)   (hifigan_causal.py:329-333)
    VocStage& s = v_st.back();
    const int T = frames * s.rate;
    cnk::ConvPostArgs a; memset(&a, 0, sizeof(a));
    if (last_merged) { a.x[0] = s.xs.ref(); a.nsrc = 1; }             // the last fused launch already stored leaky_relu(mean)
    else { for (int b = 0; b < NB; ++b) a.x[b] = s.xo[b][ND - 1].ref(); a.nsrc = NB; a.xmean = s.xs.ref(); }   // raw branch outputs: the mean is formed (and appended to xs) in the kernel
    a.slope = LR;
    a.w = ctx->vec("voc.conv_post.w"); a.bias = ctx->scalars.at("voc.conv_post.b");
    a.wav = wav_out; a.pre = pre_tanh; a.slots = d_slots; a.pos = pos; a.T = T; a.n = n; a.C = s.C; a.k = (int)ctx->scalars.at("voc.conv_post.k");
    // the step's last kernel also advances the per-slot frame counters (one launch less)
    a.adv_pos = pos_voc; a.adv_delta = frames; a.adv_ticket = cp_ticket[ws_index(st)];
    cnk::launch_conv_post(a, st);
  }
}

// ------------------------------------------------------------------------------------------------ emformer

void conan_streams::build_emformer() {
  const conan_cfg& c = ctx->cfg;
  const int D = c.emf_input_dim, Q = c.emf_segment + c.emf_right_context;
  for (int l = 0; l < c.emf_layers; ++l) {
    Ring r; r.C = D; r.rate = 1; r.L = ch::next_pow2(c.emf_left_context + c.emf_segment); r.slot_stride = (long long)r.L * D;
    r.base = alloc((size_t)max_slots * r.slot_stride); emf_state.push_back({r.base, r.slot_stride}); e_k.push_back(r);
    Ring v = r; v.base = alloc((size_t)max_slots * r.slot_stride); emf_state.push_back({v.base, v.slot_stride}); e_v.push_back(v);
  }
  // Memory bank (torchaudio max_memory_size = M > 0): every per-token buffer gets one more row for the summary token
  // (row Q; in e_x it stays zero, so the shared residual add leaves the summary's attention output untouched), the
  // normalised input is laid out [bank entries (M) | rc | utt | summary] so that the query rows [M, M+Q] and the
  // key/value rows [0, M+Q) are both contiguous; one bank ring of >= M+1 entries per layer and slot.
  const int M = c.emf_max_memory_size, QP = Q + (M > 0 ? 1 : 0);
  e_x[0] = mk_lin(QP, D); e_x[1] = mk_lin(QP, D); e_ln = mk_lin(M + QP, D); e_q = mk_lin(QP, D); e_kv = mk_lin(M + Q, 2 * D);
  e_att = mk_lin(QP, D); e_r1 = mk_lin(QP, D); e_ffn = mk_lin(QP, D); e_h = mk_lin(QP, c.emf_ffn_dim); e_r2 = mk_lin(QP, D);
  e_logits = mk_lin(c.emf_segment, c.emf_output_dim);
  if (M > 0) {
    e_bank_rows = ch::next_pow2(M + 1);
    for (int l = 0; l < c.emf_layers; ++l) {
      float* b = alloc((size_t)max_slots * e_bank_rows * D);
      emf_state.push_back({b, (long long)e_bank_rows * D});
      e_bank.push_back(b);
    }
    e_mems[0] = alloc((size_t)max_slots * D); e_mems[1] = alloc((size_t)max_slots * D);
  }
  // plan of the fused step; the per-op path below stays for shapes it does not cover (and CONAN_EMF_UNFUSED=1)
  cnk::EmfFusedArgs& a = emf_fused_args; memset(&a, 0, sizeof(a));
  if (c.emf_layers <= cnk::EMF_MAX_LAYERS) {
    for (int l = 0; l < c.emf_layers; ++l) {
      const std::string nm = "emf." + std::to_string(l);
      cnk::EmfLayerW& w = a.layers[l];
      const ch::PackedConv &q = ctx->conv(nm + ".q"), &kv = ctx->conv(nm + ".kv"), &o = ctx->conv(nm + ".out"), &f1 = ctx->conv(nm + ".ff1"), &f2 = ctx->conv(nm + ".ff2");
      if (D % 16 || c.emf_ffn_dim % 64) return;   // no fragment-major copies for such shapes: per-op path
      w.wqkv = ctx->vec(nm + ".fqkv"); w.wo = ctx->vec(nm + ".fo"); w.w1 = ctx->vec(nm + ".f1"); w.w2 = ctx->vec(nm + ".f2");
      {  // the layer's biases and LayerNorm vectors as one block (EmfLayerW::params)
        const int F = c.emf_ffn_dim;
        float* pb = alloc((size_t)11 * D + F);
        auto put = [&](int off, const float* src, int n) { HIP_CHECK(hipMemcpy(pb + off, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice)); };
        put(0, q.bias, D); put(D, kv.bias, 2 * D); put(3 * D, o.bias, D); put(4 * D, f2.bias, D);
        put(5 * D, ctx->vec(nm + ".ln_in.g"), D); put(6 * D, ctx->vec(nm + ".ln_in.b"), D);
        put(7 * D, ctx->vec(nm + ".ln_ff.g"), D); put(8 * D, ctx->vec(nm + ".ln_ff.b"), D);
        put(9 * D, ctx->vec(nm + ".ln_out.g"), D); put(10 * D, ctx->vec(nm + ".ln_out.b"), D);
        put(11 * D, f1.bias, F);
        w.params = pb;
      }
      a.kring[l] = e_k[l].base; a.vring[l] = e_v[l].base;
    }
    a.ring_slot_stride = e_k[0].slot_stride; a.lmask = e_k[0].L - 1;
    if (c.emf_output_dim != D) { a.wp = ctx->vec("emf.fproj"); a.bp = ctx->conv("emf.proj").bias; }
    a.slots = d_slots; a.past = pos_emf;
    a.L = c.emf_layers; a.R = c.emf_right_context; a.U = c.emf_segment; a.D = D; a.H = c.emf_heads; a.LC = c.emf_left_context;
    a.F = c.emf_ffn_dim; a.K = c.emf_output_dim; a.scaling = 1.0f / std::sqrt((float)(D / c.emf_heads));
    if (M > 0) {      // the one-launch step keeps the bank as projected key / value rows (the per-op plan below keeps raw entries in e_bank)
      a.M = M; a.MB = e_bank_rows; a.tanh_on_mem = c.emf_tanh_on_mem; a.bank_slot_stride = (long long)e_bank_rows * D;
      for (int l = 0; l < c.emf_layers; ++l) {
        a.bank_k[l] = alloc((size_t)max_slots * e_bank_rows * D); emf_state.push_back({a.bank_k[l], a.bank_slot_stride});
        a.bank_v[l] = alloc((size_t)max_slots * e_bank_rows * D); emf_state.push_back({a.bank_v[l], a.bank_slot_stride});
      }
    }
    { const unsigned long long per_g = (unsigned long long)std::max(c.emf_left_context, 1) * (D / 4); a.magic_per_g = (unsigned)(((1ull << 32) + per_g - 1) / per_g); }
    {  // cluster mode workspace (zeroed once: flags and epochs count up from there)
      // (the exchange buffer is indexed by cluster, and clusters of more than one workgroup exist only while groups x cs <= CUs: at most
      // CUs / 2 of them, however many slots the stream-set has - 160 KB per cluster)
      const size_t xf = cnk::emformer_cluster_xch_floats(std::min(max_slots, std::max(1, ctx->num_cu / 2)), D), fw = cnk::emformer_cluster_flag_words(max_slots);
      a.xch = alloc(xf);
      unsigned* words = reinterpret_cast<unsigned*>(alloc(fw));
      HIP_CHECK(hipMemset(words, 0, fw * sizeof(unsigned)));
      a.xflag = words; a.xepoch = words + (size_t)max_slots * cnk::EMF_MAX_LAYERS * cnk::EMF_MAX_CLUSTER;
      const char* e = dev("EMF_CLUSTER");
      emf_cluster = e ? atoi(e) : 0;       // 0: chosen per launch from the stream count
    }
    const char* off = dev("EMF_UNFUSED");
    emf_fused = cnk::emformer_fused_supported(a) && !(off && off[0] == '1');
  }
}

void conan_streams::emformer_step(int n, const float* chunk, float* out, float* logits, int32_t* codes, hipStream_t st) {
  const conan_cfg& c = ctx->cfg;
  const int D = c.emf_input_dim, R = c.emf_right_context, U = c.emf_segment, Q = R + U;
  if (emf_fused) {   // whole step in one launch (emformer_fused.hip)
    cnk::EmfFusedArgs a = emf_fused_args;
    a.chunk = chunk; a.out = out; a.logits = logits; a.codes = codes; a.n = n; a.fenced = fenced ? 1 : 0;
    a.guard = d_guard; a.fault = 0;
    // Workgroups per stream group: the step is a chain of latency-bound phases on one 16-row tile, so a few streams are
    // spread over up to 8 CUs each (feed-forward hidden units split 8 ways, one exchange per layer); with many streams
    // the groups themselves fill the CUs and the split only has to keep the launch short beside the vocoder.
    {
      const int groups = (n + cnk::emformer_fused_streams_per_block(a) - 1) / cnk::emformer_fused_streams_per_block(a);
      // The members of a cluster spin on each other's flags: every workgroup of the launch must be able to be resident at
      // once (one 129 KB workgroup per CU).  The developer override is clamped to that like the automatic choice, and a
      // CU-masked Emformer stream (CONAN_EMF_CUSTRIDE) gets no clusters at all.
      const bool masked = dev("EMF_CUSTRIDE") != nullptr && atoi(dev("EMF_CUSTRIDE")) >= 2;
      // (At most 64 workgroups - unless this launch cannot overlap anything else of the context.  A workgroup needs a whole CU for the
      // launch's 150-250 us.  In pipelined steps - the launch is on the stream-set's own Emformer stream - the vocoder's persistent
      // launches run beside it on what is left: 128 -> 64 workgroups measured 0.990 -> 0.979, 1.175 -> 1.154, 1.336 -> 1.317, 1.516 ->
      // 1.509 ms per pipelined step at 32 / 48 / 64 streams and 128 streams of 40 ms chunks; 256 cost the vocoder 6 % of the step.
      // And the limit is what keeps launches that wait inside themselves from deadlocking each other: cluster members wait for
      // partners that need WHOLE CUs, the decoder launch's 128 workgroups and the f32 pair kernel's partners wait for workgroups that
      // need PART of one - with the Emformer on more than CUs - 128 - .. workgroups two overlapping launches can each hold what the
      // other still needs (round 5: one workgroup per CU in every blocking step made tests/test_gpu_stress.py, whose blocking and
      // pipelined stream-sets overlap on the device, give up in the pair kernel's mailbox wait one run in five).  So one workgroup per
      // CU - 64 streams: 190 -> 136 us on the blocking chunk's critical path - only for a blocking step of the ONLY stream-set on its
      // device (counted over all contexts of the process; CONAN_STREAMS_SHARED_DEVICE when other processes use the GPU): its launches are serialised on one stream.  The feed-forward's sum is formed chunk by chunk in chunk order whatever the
      // cluster size (emformer_fused.hip), so the step styles and both policies produce the same bits.)
      const bool pipelined = st_emf != nullptr && st == st_emf;       // (the internal stream exists only once a pipelined step has run; a caller's null stream is not it)
      const bool alone = (!pipelined || pipe_idle) && !shared_device && live && live->load() == 1;      // (pipe_idle: a pipelined step into an EMPTY pipeline, conan_step_async)
      const int cap = (emf_cluster > 0 || alone) ? ctx->num_cu : 64;
      a.cs = emf_cluster > 0 ? std::min(emf_cluster, (int)cnk::EMF_MAX_CLUSTER) : cnk::EMF_MAX_CLUSTER;
      while (a.cs & (a.cs - 1)) a.cs &= a.cs - 1;             // a power of two
      while (a.cs > 1 && groups * a.cs > cap) a.cs >>= 1;
      if (masked && st == st_emf) a.cs = 1;
      if (test_fault == 2 && a.cs > 1) { a.fault = 0x40000000u; test_fault = 0; }      // test hook: the members wait for a flag value nobody writes
    }
    // algorithmic FLOPs of the step: per stream and layer Q = R + U query rows against the four D x D projections, the
    // D x F x 2 feed-forward, and attention over R + LC + U keys; plus the output projection
    const double Q = R + U + (c.emf_max_memory_size > 0 ? 1 : 0), Dd = D, F = c.emf_ffn_dim, keys = c.emf_max_memory_size + R + c.emf_left_context + U;
    const double fl = (double)n * (c.emf_layers * (2.0 * Q * Dd * (4.0 * Dd + 2.0 * F) + 4.0 * Q * keys * Dd) + 2.0 * U * Dd * c.emf_output_dim);
    profiled(D == 80 ? "cnk::emformer_fused_kernel<5, 10>" : "cnk::emformer_fused_kernel<4, 8>", fl, st, [&] { cnk::launch_emformer_fused(a, st); });
    return;
  }
  // token order inside the layers is [right_context | utterance] (torchaudio _EmformerLayer.infer): reorder the chunk
  const int M = c.emf_max_memory_size, QP = Q + (M > 0 ? 1 : 0);
  {
    cnk::CopyArgs ca; memset(&ca, 0, sizeof(ca));
    ca.slots = nullptr; ca.pos = nullptr; ca.lens = nullptr; ca.n = n; ca.C = D;
    ca.x = ch::lin_ref(const_cast<float*>(chunk), Q, D, U); ca.y = e_x[0].ref(0); ca.T = R; if (R > 0) cnk::launch_copy_rows(ca, st);
    ca.x = ch::lin_ref(const_cast<float*>(chunk), Q, D, 0); ca.y = e_x[0].ref(R); ca.T = U; cnk::launch_copy_rows(ca, st);
  }
  // _EmformerImpl.infer: the first layer's memory input is the mean of the raw segment
  if (M > 0) cnk::launch_emf_seg_mean(e_x[0].base, e_mems[0], n, QP, R, U, D, st);
  int cur = 0, mcur = 0;
  auto ln = [&](const TRef& x, const TRef& y, float* g, float* b, const TRef* pre) {
    cnk::LNArgs a; memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.gamma = g; a.beta = b; a.slots = nullptr; a.pos = nullptr; a.lens = nullptr; a.T = Q; a.n = n; a.C = D; a.eps = 1e-5f;
    if (pre) { a.pre = *pre; a.has_pre = 1; }
    cnk::launch_layernorm(a, st);
  };
  for (int l = 0; l < c.emf_layers; ++l) {
    const std::string nm = "emf." + std::to_string(l);
    Lin& x = e_x[cur];
    ln(x.ref(), e_ln.ref(M), ctx->vec(nm + ".ln_in.g"), ctx->vec(nm + ".ln_in.b"), nullptr);
    if (M > 0) {   // summary token, bank entries -> key/value rows, bank update with this layer's memory input
      cnk::EmfMemArgs ma; memset(&ma, 0, sizeof(ma));
      ma.ln = e_ln.base; ma.bank = e_bank[l]; ma.mems_in = e_mems[mcur]; ma.slots = d_slots; ma.past = pos_emf;
      ma.n = n; ma.R = R; ma.U = U; ma.D = D; ma.M = M; ma.MB = e_bank_rows; ma.seg = c.emf_segment;
      cnk::launch_emf_mem_prep(ma, st);
    }
    conv(mk(ctx->conv(nm + ".q"), e_ln.ref(M), e_q.ref(), n, QP, nullptr), st);          // queries: rc | utt | summary
    conv(mk(ctx->conv(nm + ".kv"), e_ln.ref(0), e_kv.ref(), n, M + Q, nullptr), st);     // keys/values: bank | rc | utt
    {
      cnk::EmfAttnArgs a; memset(&a, 0, sizeof(a));
      a.q = e_q.base; a.kv = e_kv.base; a.out = e_att.base; a.kring = e_k[l].base; a.vring = e_v[l].base;
      a.ring_slot_stride = e_k[l].slot_stride; a.slots = d_slots; a.past = pos_emf;
      a.n = n; a.R = R; a.U = U; a.D = D; a.H = c.emf_heads; a.LC = c.emf_left_context; a.lmask = e_k[l].L - 1;
      a.scaling = 1.0f / std::sqrt((float)(D / c.emf_heads));
      a.M = M; a.seg = c.emf_segment;
      cnk::launch_emf_attn(a, st);
    }
    {  // out_proj + residual with the un-normalised layer input (the summary row's residual row is zero)
      ConvArgs a = mk(ctx->conv(nm + ".out"), e_att.ref(), e_r1.ref(), n, QP, nullptr);
      a.res = x.ref(); a.has_res = 1;
      conv(a, st);
    }
    if (M > 0) {   // output_mems = clamp / tanh of the summary row: the next layer's memory input
      cnk::launch_emf_mem_out(e_r1.base, e_mems[mcur ^ 1], n, QP, Q, D, c.emf_tanh_on_mem, st);
      mcur ^= 1;
    }
    ln(e_r1.ref(), e_ffn.ref(), ctx->vec(nm + ".ln_ff.g"), ctx->vec(nm + ".ln_ff.b"), nullptr);
    { ConvArgs a = mk(ctx->conv(nm + ".ff1"), e_ffn.ref(), e_h.ref(), n, Q, nullptr); a.out_act = cnk::ACT_RELU; conv(a, st); }
    { ConvArgs a = mk(ctx->conv(nm + ".ff2"), e_h.ref(), e_r2.ref(), n, Q, nullptr); a.res = e_r1.ref(); a.has_res = 1; conv(a, st); }
    ln(e_r2.ref(), e_x[cur ^ 1].ref(), ctx->vec(nm + ".ln_out.g"), ctx->vec(nm + ".ln_out.b"), nullptr);
    cur ^= 1;
  }
  if (out) {
    cnk::CopyArgs ca; memset(&ca, 0, sizeof(ca));
    ca.n = n; ca.C = D; ca.T = U; ca.x = e_x[cur].ref(R); ca.y = ch::lin_ref(out, U, D);
    cnk::launch_copy_rows(ca, st);
  }
  if (logits || codes) {
    float* lg = logits ? logits : e_logits.base;
    const int K = c.emf_output_dim;
    if (c.emf_output_dim != D) conv(mk(ctx->conv("emf.proj"), e_x[cur].ref(R), ch::lin_ref(lg, U, K), n, U, nullptr), st);
    else { cnk::CopyArgs ca; memset(&ca, 0, sizeof(ca)); ca.n = n; ca.C = D; ca.T = U; ca.x = e_x[cur].ref(R); ca.y = ch::lin_ref(lg, U, D); cnk::launch_copy_rows(ca, st); }
    if (codes) { cnk::ArgmaxArgs a; a.x = lg; a.idx = codes; a.rows = n * U; a.C = K; cnk::launch_argmax(a, st); }
  }
  cnk::launch_advance(pos_emf, d_slots, n, U, st);
}

// ------------------------------------------------------------------------------------------------ conan decoder

void conan_streams::build_decoder() {
  const conan_cfg& c = ctx->cfg;
  const int H = c.hidden_size, F = max_frames;
  c_emb = mk_ring(H, 1, c.content_kernel - 1, &dec_state);
  c_pin2 = mk_ring(H, 1, c.predictor_kernel - 1, &dec_state);
  // (uv predictor depth / width and the aligner's feed-forward width come from the checkpoint's tensors, ctx.hip finalize_conan)
  const int n_uv = (int)ctx->scalars.at("conan.uv.n"), uvh = (int)ctx->scalars.at("conan.uv.hidden"), ffn = (int)ctx->scalars.at("conan.align.ffn");
  for (int i = 0; i + 1 < n_uv; ++i) c_uvh.push_back(mk_ring(ctx->conv("conan.uv." + std::to_string(i)).Cout, 1, c.predictor_kernel - 1, &dec_state));
  // one post-LN ring per (block, sub-layer): each layer keeps its own left context
  for (int b = 0; b < c.dec_num_blocks; ++b)
    for (int j = 0; j < c.dec_layers_in_block; ++j) c_lnrs.push_back(mk_ring(H, 1, (c.dec_kernel - 1) * c.dec_dilations[b], &dec_state));
  c_lastr = mk_ring(H, 1, c.dec_post_kernel - 1, &dec_state);
  c_pin = mk_lin(F, H); c_q = mk_lin(F, H); c_att = mk_lin(F, H); c_a1 = mk_lin(F, H); c_a2 = mk_lin(F, H);
  c_ff = mk_lin(F, ffn); c_uv5 = mk_lin(F, uvh); c_x[0] = mk_lin(F, H); c_x[1] = mk_lin(F, H); c_h = mk_lin(F, 2 * H);
  c_post = mk_lin(F, H); c_mask_blk = mk_lin(F, 1); c_mask_blk2 = mk_lin(F, 1); c_mask_out = mk_lin(F, 1); c_mel = mk_lin(F, c.num_mels);
  c_part = mk_lin(F, 32 * H);       // decoder megakernel: the group members' (8; xcd mode: 32 virtual ones) partial sums of a fused feed-forward, [member][row][H]
  c_part2 = mk_lin(F, 16 * H);      // ... and a second set: consecutive fused conv blocks write one while the members still read the other
  S_max = (max_ref + 3) / 4;
  if (S_max > 512) throw Error(CONAN_ERR_UNSUPPORTED, "max_ref_frames > 2048");
  c_style = alloc((size_t)max_slots * H);
  c_kv = alloc((size_t)max_slots * 2 * S_max * 2 * H);
  c_kmask = alloc((size_t)max_slots * S_max);
  c_slen = (int*)alloc((size_t)max_slots);
  c_vqids = (int*)alloc((size_t)max_slots * S_max);
  // style-pass workspace
  sp_batch = std::min(max_slots, 8);
  const int TR = max_ref + 2 * PADR, SR = S_max + 2 * PADR;
  s_mel = mk_lin(TR, c.num_mels, sp_batch); s_np = mk_lin(TR, 1, sp_batch); s_wnm = mk_lin(TR, 1, sp_batch);
  s_x[0] = mk_lin(TR, H, sp_batch); s_x[1] = mk_lin(TR, H, sp_batch); s_ln = mk_lin(TR, H, sp_batch); s_h = mk_lin(TR, 2 * H, sp_batch);
  s_blkm = mk_lin(TR, 1, sp_batch);
  const int NM = c.num_mels;      // WN hidden width = encoder width = mel bins
  s_wx = mk_lin(TR, NM, sp_batch); s_wout = mk_lin(TR, NM, sp_batch); s_win = mk_lin(TR, 2 * NM, sp_batch); s_acts = mk_lin(TR, NM, sp_batch);
  s_rs = mk_lin(TR, 2 * NM, sp_batch);
  s_ph = mk_lin(SR, NM, sp_batch); s_pm = mk_lin(SR, 1, sp_batch); s_px[0] = mk_lin(SR, NM, sp_batch); s_px[1] = mk_lin(SR, NM, sp_batch);
  s_pln = mk_lin(SR, NM, sp_batch); s_phh = mk_lin(SR, 2 * NM, sp_batch); s_pblk = mk_lin(SR, 1, sp_batch);
  s_enc = mk_lin(SR, H, sp_batch); s_dots = mk_lin(SR, c.nvq, sp_batch); s_cat = mk_lin(SR, 2 * H, sp_batch); s_tok = mk_lin(SR, H, sp_batch);
  s_kvtmp = mk_lin(SR, 2 * H, sp_batch);
  s_ids = (int*)alloc((size_t)sp_batch * S_max);
}
