"""GPU tests added in round 2: the Emformer pinned against the public torchaudio package where the box has it, the
reference-generated loop fixture through the HIP steps, BASELINE configs[4] as specified (B=128, seg 2 + rc 2,
320 ms window), per-stage vocoder taps and VQ ids against the goldens, the batch runner against the oracle loop, the
pipelined/blocking equivalence at full size (split-K on both internal streams), and the reference-shaped seams
(StreamingVoiceConversion(hp) from checkpoints, Conan.forward(spk_embed=), EmformerDistillModel.inference)."""
import numpy as np
import pytest
import torch

from conan_amd import configs, synth
from tests.conftest import ARITHS, assert_arith_ran, kernels_of, load_golden

pytestmark = pytest.mark.gpu


def _t(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def _ctx(chp=None, vhp=None, tiny=False, emformer=True, conan=True, hifigan=True):
    from conan_amd.runtime import Context
    chp = chp or configs.conan_hparams(tiny)
    vhp = vhp or configs.hifigan_hparams(tiny)
    ctx = Context(chp if (emformer or conan) else None, vhp if hifigan else None, 0, emformer, conan, hifigan)
    if emformer:
        ctx.load_state_dict("emformer", synth.emformer_state_dict(chp, 0))
    if conan:
        ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0))
    if hifigan:
        ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
    ctx.finalize()
    return ctx, chp, vhp


# ---------------------------------------------------------------------------------------------- Emformer pin
@pytest.mark.parametrize("variant", ["rc2_seg4", "rc0_seg4", "rc2_seg2", "rc2_seg4_mem4"])
def test_emformer_against_torchaudio(variant):
    """The Emformer arithmetic is third-party (torchaudio==2.5.1, requirements.txt:3; call sites
    modules/Emformer/emformer.py:14-22, inference/Conan.py:115-124).  Where the public torchaudio package is installed,
    torchaudio.models.Emformer loads the same state_dict and BOTH the oracle restatement and the HIP step are compared
    with its .infer() over 18+ chunks (the 50-frame left-context cache fills and wraps)."""
    try:
        import torchaudio
    except Exception as e:  # noqa: BLE001  (ImportError, or an ABI mismatch with the installed torch)
        print(f"\n[emformer-pin] SKIPPED: torchaudio is not importable on this box ({type(e).__name__}: {e})")
        pytest.skip("torchaudio not importable: Emformer parity stays unpinned on this box")
    from oracle import emformer as oemf
    from oracle.common import to_torch_sd
    rc = 0 if variant.startswith("rc0") else 2
    seg = 2 if "seg2" in variant else 4
    mem = 4 if variant.endswith("mem4") else 0
    chp = dict(configs.conan_hparams(), right_context=rc, chunk_size=20 * seg, emformer_max_memory_size=mem)
    ctx, chp, _ = _ctx(chp, conan=False, hifigan=False)
    sd_np = synth.emformer_state_dict(chp, 0)
    sd = to_torch_sd(sd_np)
    cfg = oemf.EmformerCfg(chp)
    em = torchaudio.models.Emformer(80, 8, 2048, chp["emformer_layers"], seg, left_context_length=50, right_context_length=rc,
                                    max_memory_size=mem)
    em.load_state_dict({k[len("emformer."):]: v for k, v in sd.items() if k.startswith("emformer.")}, strict=True)
    em.eval()
    B, T = 3, 18 * seg + 3           # ragged tail: the last chunk is padded by repeating the last frame
    mel = torch.from_numpy(synth.mel(T, 21, B))
    st = ctx.streams(B, max_frames=seg, max_ref_frames=16)
    slots = [1, 2, 0]
    st.reset(slots)
    s_ta = s_or = None
    worst_or = worst_hip = 0.0
    for pos, emit, chunk in oemf.chunk_iter(mel, seg, rc):
        lengths = torch.full((B,), chunk.shape[1], dtype=torch.long)
        with torch.no_grad():
            o_ta, _, s_ta = em.infer(chunk, lengths, s_ta)
        o_or, _, s_or = oemf.emformer_infer(sd, cfg, chunk, lengths, s_or)
        o_hip, _, _ = st.emformer_step(slots, chunk.cuda())
        worst_or = max(worst_or, float((o_or - o_ta).abs().max()))
        worst_hip = max(worst_hip, float((o_hip.cpu() - o_ta).abs().max()))
        np.testing.assert_allclose(o_or.numpy(), o_ta.numpy(), atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(o_hip.cpu().numpy(), o_ta.numpy(), atol=1e-4, rtol=1e-4)
    print(f"\n[emformer-pin] RAN against torchaudio {torchaudio.__version__} ({variant}): oracle max|d| {worst_or:.2e}, HIP max|d| {worst_hip:.2e}")
    st.close(); ctx.close()


@pytest.mark.parametrize("plan", ["fused", "per-op"])
@pytest.mark.parametrize("M,tanh", [(4, False), (2, True)])
def test_emformer_memory_bank_vs_oracle(M, tanh, plan):
    """The memory bank of torchaudio's Emformer (max_memory_size > 0: summary token, per-layer bank of the last M segment
    memories as extra attention keys, clamp / tanh on the produced memory) - BASELINE.json north_star "memory-bank
    update".  modules/Emformer/emformer.py:14-22 never enables it, so this is an extra, non-parity datapoint: the HIP
    step (one launch, or the per-op plan) against the oracle restatement (itself checked against a whole-sequence formulation on the CPU) over
    20 chunks: bank ramp-up, saturation and roll-over, left-context wrap, a stream restarted half way."""
    from oracle import emformer as oemf
    from oracle.common import to_torch_sd
    # plan "fused": the one-launch step (round 3: summary and memory-input rows in the 16-row tile, the bank as projected key /
    # value rows at the head of the key tables); "per-op": the separate kernels (CONAN_EMF_UNFUSED=1, read at stream-set creation)
    dev_plan = "EMF_UNFUSED=1" if plan == "per-op" else None
    chp = dict(configs.conan_hparams(), emformer_max_memory_size=M, emformer_tanh_on_mem=tanh)
    ctx, chp, _ = _ctx(chp, conan=False, hifigan=False)
    assert ctx.cfg.emf_max_memory_size == M
    sd = to_torch_sd(synth.emformer_state_dict(chp, 0))
    cfg = oemf.EmformerCfg(chp)
    B, T = 3, 80
    mel = torch.from_numpy(synth.mel(T, 33, B))
    st = ctx.streams(4, max_frames=4, max_ref_frames=16, dev_plan=dev_plan)
    slots = [3, 0, 2]
    st.reset(slots)
    state = None
    for n_chunk, (pos, emit, chunk) in enumerate(oemf.chunk_iter(mel, 4, 2)):
        if n_chunk == 11:      # all three streams restart (the oracle's batch shares one past_length): state and bank cleared
            st.reset(slots, which=1)
            state = None
        lengths = torch.full((B,), 6, dtype=torch.long)
        o_ref, _, state = oemf.emformer_infer(sd, cfg, chunk, lengths, state)
        lg_ref, codes_ref = oemf.logits_and_codes(sd, o_ref)
        o, lg, codes = st.emformer_step(slots, chunk.cuda())
        np.testing.assert_allclose(o.cpu().numpy(), o_ref.numpy(), atol=1e-4, rtol=1e-4)
        np.testing.assert_allclose(lg.cpu().numpy(), lg_ref.numpy(), atol=2e-4, rtol=1e-4)
        top2 = lg_ref.topk(2, -1).values
        safe = (top2[..., 0] - top2[..., 1]) > 1e-3
        assert torch.equal(codes.cpu().long()[safe], codes_ref[safe])
    # which kernels ran: the one-launch step, or the per-op plan's attention kernel
    st.profile_begin()
    st.emformer_step(slots, chunk.cuda())
    st.profile_end()
    names = [k[0] for k in st.profile_kernels()]
    assert any("emformer_fused_kernel" in k for k in names) == (plan == "fused"), names
    st.close(); ctx.close()


def test_emformer_cluster_mode_matches_single_workgroup():
    """The fused Emformer step spread over clusters of 8 / 4 / 2 workgroups per stream group (feed-forward hidden units
    split over the cluster, one write-through exchange of partial sums per layer, emformer_fused.hip) against the same
    step with one workgroup per group, over 30 chunks (left-context ring wraps, a stream restarts half way).  Round 5: every hidden
    chunk's partial sum is formed from zero and the chunks are added in chunk order whatever the cluster size, so outputs, logits and
    codes are BIT-IDENTICAL across cluster sizes (until then: fp32 re-association noise, 1e-5) - which is what lets blocking steps
    take one workgroup per CU while pipelined steps keep 64.  Stream sets read CONAN_EMF_CLUSTER when they are created, so one process
    can hold all variants."""
    from oracle import emformer as oemf
    ctx, chp, _ = _ctx(conan=False, hifigan=False)
    B, T = 5, 120
    mel = torch.from_numpy(synth.mel(T, 41, B))
    slots = [4, 1, 0, 3, 2]
    sets = {}
    for cs in ("1", "2", "4", "8", "8"):
        sets.setdefault(cs, []).append(ctx.streams(6, max_frames=4, max_ref_frames=16, dev_plan="EMF_CLUSTER=" + cs))
    allsets = [st for v in sets.values() for st in v]
    for st in allsets:
        st.reset(slots)
    for n_chunk, (pos, emit, chunk) in enumerate(oemf.chunk_iter(mel, 4, 2)):
        if n_chunk == 13:
            for st in allsets:
                st.reset([slots[1]], which=1)
        x = chunk.cuda()
        o1, lg1, c1 = sets["1"][0].emformer_step(slots, x)
        for cs in ("2", "4", "8"):
            o, lg, c = sets[cs][0].emformer_step(slots, x)
            assert torch.equal(lg, lg1) and torch.equal(o, o1) and torch.equal(c, c1), f"cluster size {cs} differs from one workgroup per group"

        o8b, lg8b, c8b = sets["8"][1].emformer_step(slots, x)
        assert torch.equal(lg8b, lg) and torch.equal(c8b, c) and torch.equal(o8b, o)      # same cluster size: same bits
    for st in allsets:
        st.close()
    ctx.close()


def test_merged_branch_launches_equal_separate_branches_bitwise():
    """Last dilation of a vocoder stage: the merged-branch build of the fused ResBlock pass (one workgroup runs the three
    branches of a (slot, row tile) group and stores leaky_relu(mean); 64 streams: stages C = 64 and C = 32) against
    separate branch tiles + mean_act / conv_post forming the mean - the same 64 streams, the same other kernels, so the
    audio must be identical bit for bit.  A third stream-set alternates between 64-slot steps (merged) and two 32-slot
    steps (fewer groups than CUs: not merged; its other kernels differ, so 1e-5): the mean ring stays valid whichever
    way the previous step produced it."""
    ctx, _, vhp = _ctx(emformer=False, conan=False)
    B, steps = 64, 4
    mel = torch.from_numpy(synth.mel(4 * steps, 77, B)).cuda()                    # [B, 16, 80]
    merged = ctx.streams(B, max_frames=4, max_ref_frames=16)
    alt = ctx.streams(B, max_frames=4, max_ref_frames=16)
    separate = ctx.streams(B, max_frames=4, max_ref_frames=16, dev_plan="RB_NOMERGE=1")
    slots = list(range(B))
    for st in (merged, separate, alt):
        st.reset(slots)
    for t in range(steps):
        chunk = mel[:, 4 * t:4 * t + 4]
        wa = merged.hifigan_step(slots, chunk)
        wb = separate.hifigan_step(slots, chunk)
        assert torch.equal(wa, wb), float((wa - wb).abs().max())
        if t % 2 == 0:
            wc = alt.hifigan_step(slots, chunk)
        else:
            wc = torch.cat([alt.hifigan_step(slots[:32], chunk[:32]), alt.hifigan_step(slots[32:], chunk[32:])])
        assert float((wa - wc).abs().max()) < 1e-5
    for st in (merged, separate, alt):
        st.close()
    ctx.close()


# ---------------------------------------------------------------------------------------------- loop fixture
@pytest.mark.parametrize("tag,tiny,arith", [("tiny", True, "f32"), ("full", False, "f32"), ("full", False, "limb")])
def test_loop_golden_through_hip_steps(tag, tiny, arith):
    """tests/golden/loop_{tiny,full}.npz: mel / wav of the reference-semantics loop (inference/Conan.py:95-156) built
    from the IMPORTED reference modules for a given code sequence (tools/make_goldens.py).  The HIP path is fed the
    golden codes chunk by chunk (conan_decoder_step -> conan_hifigan_step, 4 frames per step)."""
    g = load_golden(f"loop_{tag}.npz")
    ctx, chp, vhp = _ctx(tiny=tiny, emformer=False)
    codes = torch.from_numpy(g["codes"]).int().cuda()[None]
    T = codes.shape[1]
    st = ctx.streams(2, max_frames=4, max_ref_frames=64, arith=arith)
    st.reset([1])
    st.set_reference([1], torch.from_numpy(g["ref"]).cuda())
    mels, wavs = [], []
    for p in range(0, T, 4):
        m = st.decoder_step([1], codes[:, p:p + 4])
        mels.append(m)
        wavs.append(st.hifigan_step([1], m))
    mel, wav = torch.cat(mels, 1)[0].cpu().numpy(), torch.cat(wavs, 1)[0].cpu().numpy()
    np.testing.assert_allclose(mel, g["mel"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(wav, g["wav"], atol=1e-4, rtol=0)
    if not tiny:
        assert_arith_ran(kernels_of(st, lambda: st.hifigan_step([1], mels[0])), arith)
    st.close(); ctx.close()


# ---------------------------------------------------------------------------------------------- vocoder / VQ taps
@pytest.mark.parametrize("tag,tiny,arith", [("tiny", True, "f32"), ("full", False, "f32"), ("full", False, "limb")])
def test_vocoder_stage_taps_match_reference_goldens(tag, tiny, arith):
    """Per-stage tensors of HifiGanGenerator.forward captured by forward hooks on the imported reference
    (conv_pre_12, ups.{i}_12, pre_tanh_12) against the taps of conan_hifigan_step_taps, streamed 4 frames per step."""
    g = load_golden(f"hifigan_{tag}.npz")
    ctx, _, vhp = _ctx(tiny=tiny, emformer=False, conan=False)
    mel = torch.from_numpy(g["mel_12"]).transpose(1, 2).contiguous().cuda()       # [1,12,80]
    st = ctx.streams(1, max_frames=4, max_ref_frames=16, arith=arith)
    st.reset([0])
    parts = [st.hifigan_step_taps([0], mel[:, p:p + 4]) for p in range(0, 12, 4)]
    wav = torch.cat([p[0] for p in parts], 1)[0].cpu().numpy()
    pre = torch.cat([p[1] for p in parts], 1)[0].cpu().numpy()
    cpre = torch.cat([p[2] for p in parts], 1)[0].cpu().numpy()                   # [12, C0]
    np.testing.assert_allclose(wav, g["wav_12"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(pre, g["pre_tanh_12"][0], atol=1e-4 * max(1.0, np.abs(g["pre_tanh_12"]).max()), rtol=0)
    ref_cpre = g["conv_pre_12"].T
    np.testing.assert_allclose(cpre, np.where(ref_cpre > 0, ref_cpre, 0.1 * ref_cpre), atol=1e-4, rtol=1e-4)
    for i in range(len(vhp["upsample_rates"])):
        up = torch.cat([p[3][i] for p in parts], 1)[0].cpu().numpy()              # [12*rate_i, C_i]
        ref = g[f"ups.{i}_12"].T
        assert up.shape == ref.shape
        np.testing.assert_allclose(up, ref, atol=1e-4 * max(1.0, np.abs(ref).max()), rtol=0)
    if not tiny:
        assert_arith_ran(kernels_of(st, lambda: st.hifigan_step([0], mel[:, :4])), arith)
    st.close(); ctx.close()


@pytest.mark.parametrize("tag,tiny", [("tiny", True), ("full", False)])
def test_vq_ids_match_reference_golden(tag, tiny):
    """VQEmbeddingEMA.encode argmin (prosody_util.py:34-46) of the style pass: integer intermediate, exact."""
    g = load_golden(f"conan_{tag}.npz")
    ctx, _, _ = _ctx(tiny=tiny, emformer=False, hifigan=False)
    st = ctx.streams(3, max_frames=4, max_ref_frames=160)
    st.set_reference([2], torch.from_numpy(g["ref"]).cuda())
    ids, cnt = st.prosody_ids([2])
    n = int(cnt[0])
    assert n == g["vq_ids"].shape[1]
    assert np.array_equal(ids[0, :n].cpu().numpy(), g["vq_ids"][0])
    assert bool((ids[0, n:] == -1).all())
    # a slot without a reference is refused, slot by slot (not one flag per stream-set)
    from conan_amd._lib import ConanError
    with pytest.raises(ConanError):
        st.decoder_step([0], torch.zeros(1, 4, dtype=torch.int32, device="cuda"))
    with pytest.raises(ConanError):
        st.prosody_ids([1])
    st.close(); ctx.close()


# ---------------------------------------------------------------------------------------------- configs[4]
@pytest.mark.parametrize("arith", ARITHS)
def test_config4_b128_seg2_windowed_320ms(arith):
    """BASELINE configs[4] as specified: 40 ms chunks (seg 2 + rc 2), batch = 128 streams, Conan / vocoder in windowed
    mode with a 320 ms (16-frame) context: state reset + 18 frames per step, last 2 frames kept; the Emformer stays
    stateful.  Oracle for three streams: the reference modules' restatement fed the same 18-frame window;
    every other stream through slot independence (same stream alone == inside the batch of 128)."""
    from conan_amd.engine import StreamingVoiceConversionEngine
    from oracle import conan as oconan
    from oracle import hifigan as ohifi
    from oracle.common import to_torch_sd
    chp = dict(configs.conan_hparams(), chunk_size=40)
    ctx, chp, vhp = _ctx(chp)
    assert ctx.cfg.emf_segment == 2 and ctx.cfg.emf_right_context == 2
    B, CTX, SEG, Tr = 128, 16, 2, 40
    nchunks = CTX // SEG + 3
    T = nchunks * SEG + 2
    src = torch.from_numpy(np.concatenate([synth.mel(T, 1234 + s) for s in range(B)])).cuda()
    ref = torch.from_numpy(np.concatenate([synth.mel(Tr, 4321 + s) for s in range(B)])).cuda()
    eng = StreamingVoiceConversionEngine(ctx, B, max_ref_frames=64, max_frames=CTX + SEG, arith=arith)
    assert eng.st.arith == arith
    eng.start(ref)
    hist = torch.zeros(B, 0, dtype=torch.int32, device="cuda")
    outs = []
    for k in range(nchunks):
        chunk = src[:, k * SEG:k * SEG + SEG + 2].contiguous()
        codes, wav, mel = eng.windowed_step(chunk, hist[:, -CTX:], return_mel=True)
        hist = torch.cat([hist, codes], 1)
        outs.append((wav, mel))
    assert hist.shape == (B, nchunks * SEG)
    csd, vsd = to_torch_sd(synth.conan_state_dict(chp, 0)), to_torch_sd(synth.hifigan_state_dict(vhp, 0))
    k = nchunks - 1                                                   # a step with the full 16-frame context
    win = hist[:, k * SEG + SEG - (CTX + SEG):k * SEG + SEG]
    assert win.shape[1] == CTX + SEG
    wav_k, mel_k = outs[k]
    for b in (0, 77, 127):
        with torch.no_grad():
            m_ref = oconan.conan_forward(csd, chp, win[b:b + 1].cpu().long(), ref[b:b + 1].cpu())["mel_out"]     # [1,18,80]
            w_ref = ohifi.generator_forward(vsd, vhp, m_ref.transpose(1, 2)).view(-1)
        np.testing.assert_allclose(mel_k[b].cpu().numpy(), m_ref[0, -SEG:].numpy(), atol=1e-4, rtol=1e-4)
        np.testing.assert_allclose(wav_k[b].cpu().numpy(), w_ref[-SEG * 320:].numpy(), atol=1e-4, rtol=0)
    # slot independence: streams run alone (batch of 2, other slots) reproduce their rows of the batch of 128
    solo = StreamingVoiceConversionEngine(ctx, 2, max_ref_frames=64, max_frames=CTX + SEG, arith=arith)
    for pair in ([5, 100], [63, 64]):
        solo.start(ref[pair])
        h = torch.zeros(2, 0, dtype=torch.int32, device="cuda")
        for kk in range(nchunks):
            c2, w2, m2 = solo.windowed_step(src[pair, kk * SEG:kk * SEG + SEG + 2].contiguous(), h[:, -CTX:], return_mel=True)
            h = torch.cat([h, c2], 1)
            np.testing.assert_allclose(m2.cpu().numpy(), outs[kk][1][pair].cpu().numpy(), atol=2e-5, rtol=1e-5)
            np.testing.assert_allclose(w2.cpu().numpy(), outs[kk][0][pair].cpu().numpy(), atol=2e-5, rtol=0)
        assert torch.equal(h, hist[pair])
    wmel = torch.from_numpy(synth.mel(CTX + SEG, 3, B)).cuda()
    eng.st.reset(eng.slots, which=4)
    assert_arith_ran(kernels_of(eng.st, lambda: eng.st.hifigan_step(eng.slots, wmel)), arith)
    eng.st.close(); solo.st.close(); ctx.close()


# ---------------------------------------------------------------------------------------------- pipelined == blocking
@pytest.mark.parametrize("B", [1, 4])
def test_pipelined_equals_blocking_full_size_small_batch(B):
    """Full-size models at B = 1 and 4: both the front-end (aligner FFN, K = 2048) and the vocoder (ups.0, K = 8192)
    use inter-block split-K there, and conan_step_async runs them concurrently on two internal streams - each with its
    own split-K workspace.  38+ pipelined chunks against the blocking conan_step, several rounds, bit for bit."""
    ctx, chp, vhp = _ctx()
    T, Tr = 160, 48
    src = torch.from_numpy(np.concatenate([synth.mel(T, 700 + s) for s in range(B)])).cuda()
    ref = torch.from_numpy(np.concatenate([synth.mel(Tr, 800 + s) for s in range(B)])).cuda()
    slots = list(range(B))
    a, b = ctx.streams(B, 4, 64), ctx.streams(B, 4, 64)
    hop = ctx.hop
    chunks = [src[:, p:p + 6].contiguous() for p in range(0, T - 6, 4)]
    assert len(chunks) >= 38
    for rnd in range(3):
        for st in (a, b):
            st.reset(slots)
            st.set_reference(slots, ref)
        outs_a = [tuple(x.clone() for x in a.step(slots, c)) for c in chunks]
        outs_b = []
        for c in chunks:
            cb = torch.empty(B, 4, dtype=torch.int32, device="cuda"); mb = torch.empty(B, 4, 80, device="cuda"); wb = torch.empty(B, 4 * hop, device="cuda")
            b.step_async(slots, c, wb, codes=cb, mel_out=mb)
            outs_b.append((cb, mb, wb))
        b.join()
        torch.cuda.synchronize()
        for k, ((ca, ma, wa), (cb, mb, wb)) in enumerate(zip(outs_a, outs_b)):
            assert torch.equal(ca, cb) and torch.equal(ma, mb) and torch.equal(wa, wb), f"round {rnd} chunk {k}"
    a.close(); b.close(); ctx.close()


# ---------------------------------------------------------------------------------------------- runner vs oracle
def test_batch_file_runner_against_oracle_loop(tmp_path):
    """VoiceConversionRunner output files against the ORACLE chunk loop run on the same mels (the GPU front-end's mel
    of each file is handed to the oracle, so the comparison isolates the path after the front-end): int16 samples
    within 4 LSB (1e-4 of full scale = 3.3 LSB after the 32767 scaling)."""
    import json
    from scipy.io import wavfile
    from conan_amd.inference.Conan import StreamingVoiceConversion
    from conan_amd.inference.run_voice_conversion import VoiceConversionRunner
    from conan_amd.utils.audio.io import save_wav
    from oracle import emformer as oemf
    from oracle import loop as oloop
    from oracle.common import to_torch_sd
    chp, vhp = configs.conan_hparams(True), configs.hifigan_hparams(True)
    sds_np = {"emformer": synth.emformer_state_dict(chp, 0), "conan": synth.conan_state_dict(chp, 0), "hifigan": synth.hifigan_state_dict(vhp, 0)}
    sds = {k: _t(v) for k, v in sds_np.items()}
    tsd = {k: to_torch_sd(v) for k, v in sds_np.items()}
    sr = 16000
    rng = np.random.default_rng(11)
    pairs = []
    for k, (ds, dr) in enumerate(((0.45, 0.40), (0.31, 0.52))):
        ts, tr = np.arange(int(ds * sr)) / sr, np.arange(int(dr * sr)) / sr
        s = 0.4 * np.sin(2 * np.pi * (210 + 70 * k) * ts) + 0.02 * rng.standard_normal(ts.shape)
        r = 0.3 * np.sin(2 * np.pi * (140 + 50 * k) * tr) * np.cos(2 * np.pi * 2 * tr)
        save_wav(s, str(tmp_path / f"s{k}.wav"), sr); save_wav(r, str(tmp_path / f"r{k}.wav"), sr)
        pairs.append({"src_wav": str(tmp_path / f"s{k}.wav"), "ref_wav": str(tmp_path / f"r{k}.wav"), "output_name": f"out{k}.wav"})
    cfgp = tmp_path / "pairs.json"
    cfgp.write_text(json.dumps({"total_pairs": len(pairs), "conversion_pairs": pairs}))
    runner = VoiceConversionRunner(str(cfgp), chp, vhp, sds, output_dir=str(tmp_path / "out"), streams=2)
    res = runner.run_all_conversions()
    assert res["successful"] == 2 and res["failed"] == 0
    vc = StreamingVoiceConversion(chp, vhp, sds)
    cfg = oemf.EmformerCfg(chp)
    for k, p in enumerate(pairs):
        m_src, m_ref = vc._wav_to_mel(p["src_wav"]).cpu().numpy(), vc._wav_to_mel(p["ref_wav"]).cpu().numpy()
        w_ref, _, _ = oloop.infer_once_stateful(tsd["emformer"], cfg, tsd["conan"], chp, tsd["hifigan"], vhp, m_src, m_ref)
        got = wavfile.read(str(tmp_path / "out" / f"out{k}.wav"))[1].astype(np.int32)
        want = (np.asarray(w_ref, dtype=np.float32) * 32767).astype(np.int16).astype(np.int32)   # utils/audio/io.py:7-13
        assert len(got) == len(want)
        assert np.abs(got - want).max() <= 4          # 1e-4 of full scale = 3.3 LSB


# ---------------------------------------------------------------------------------------------- reference-shaped seams
def test_streaming_voice_conversion_from_checkpoints(tmp_path):
    """StreamingVoiceConversion(hp) with no in-memory weights, as inference/Conan.py:26-55 constructs it: the three
    models come from hp['work_dir'] / hp['vocoder_ckpt'] (config.yaml + 'model_gen') / hp['emformer_ckpt'] through
    load_ckpt(strict=False); the result equals the object built from the same state_dicts in memory."""
    import yaml
    from conan_amd.inference.Conan import StreamingVoiceConversion
    chp, vhp = configs.conan_hparams(True), configs.hifigan_hparams(True)
    sds = {"emformer": _t(synth.emformer_state_dict(chp, 3)), "conan": _t(synth.conan_state_dict(chp, 3)), "hifigan": _t(synth.hifigan_state_dict(vhp, 3))}
    wd, vd, ed = tmp_path / "conan", tmp_path / "hifigan_vc", tmp_path / "emformer"
    for d in (wd, vd, ed):
        d.mkdir()
    torch.save({"state_dict": {"model": sds["conan"]}, "global_step": 100}, str(wd / "model_ckpt_steps_100.ckpt"))
    torch.save({"state_dict": {"model": {k: v * 0 for k, v in sds["conan"].items()}}}, str(wd / "model_ckpt_steps_50.ckpt"))   # older: must be ignored
    torch.save({"state_dict": {"model_gen": sds["hifigan"], "model_disc": {}}}, str(vd / "model_ckpt_steps_7.ckpt"))
    (vd / "config.yaml").write_text(yaml.safe_dump(vhp))
    torch.save({"state_dict": {"model": sds["emformer"]}}, str(ed / "model_ckpt_steps_9.ckpt"))
    hp = dict(chp, work_dir=str(wd), vocoder_ckpt=str(vd), emformer_ckpt=str(ed))
    vc_files = StreamingVoiceConversion(hp)
    vc_mem = StreamingVoiceConversion(chp, vhp, sds)
    inp = {"ref_mel": synth.mel(30, 2)[0], "src_mel": synth.mel(22, 1)[0]}
    wa, ma = vc_files.infer_once(inp)
    wb, mb = vc_mem.infer_once(inp)
    assert wa.shape == (22 * 320,) and np.array_equal(wa, wb) and np.array_equal(ma, mb)
    with pytest.raises(AssertionError):
        StreamingVoiceConversion(dict(hp, work_dir=str(tmp_path / "nowhere")))
    with pytest.raises(ValueError):
        StreamingVoiceConversion(dict(hp, vocoder="NoSuchVocoder"))


def test_conan_forward_with_spk_embed():
    """Conan.forward(spk_embed=...) (modules/Conan/Conan.py:146-149): the given vector replaces encode_spk_embed's; the
    prosody still comes from ref.  Oracle: decode_frames with the cache's style vector replaced."""
    from conan_amd.modules.Conan.Conan import Conan
    from oracle import conan as oconan
    from oracle.common import to_torch_sd
    chp = configs.conan_hparams(True)
    sd_np = synth.conan_state_dict(chp, 0)
    m = Conan(0, chp)
    m.load_state_dict(_t(sd_np), strict=True)
    g = load_golden("conan_tiny.npz")
    content, ref = torch.from_numpy(g["content"][:, :24]), torch.from_numpy(g["ref"])
    spk = torch.from_numpy(np.random.default_rng(3).standard_normal((1, 1, chp["hidden_size"])).astype(np.float32) * 0.3)
    ret = m(content=content.cuda(), spk_embed=spk.cuda(), ref=ref.cuda(), infer=True)
    np.testing.assert_allclose(ret["style_embed"].cpu().numpy(), spk.numpy(), atol=0, rtol=0)
    sd = to_torch_sd(sd_np)
    with torch.no_grad():
        cache = oconan.style_pass(sd, chp, ref)
        cache["style_embed"] = spk
        want = oconan.decode_frames(sd, chp, content.long(), cache, {})["mel_out"]
    np.testing.assert_allclose(ret["mel_out"].cpu().numpy(), want.numpy(), atol=1e-4, rtol=1e-4)
    with pytest.raises(ValueError):
        m(content=content.cuda(), infer=True)


def test_emformer_inference_returns_both_heads():
    """EmformerDistillModel.inference (modules/Emformer/emformer.py:48-98): proj(features), or the tuple
    (proj1(features), proj2(features)) when mode == 'both'."""
    from conan_amd.modules.Emformer.emformer import EmformerDistillModel
    from oracle import emformer as oemf
    from oracle.common import to_torch_sd
    chp = dict(configs.conan_hparams(True), mode="both", emformer_output_dim=768)
    sd_np = synth.emformer_state_dict(chp, 0, output_dim=768)
    model = EmformerDistillModel(chp, output_dim=768)
    model.load_state_dict(_t(sd_np), strict=True)
    sd = to_torch_sd(sd_np)
    cfg = oemf.EmformerCfg(chp)
    mel = torch.from_numpy(synth.mel(19, 5, 2))
    o1, o2 = model.inference(mel.cuda())
    assert o1.shape == (2, 19, 100) and o2.shape == (2, 19, 768)
    state, feats = None, []
    for pos, emit, chunk in oemf.chunk_iter(mel, cfg.segment_length, cfg.right_context_length):
        o, _, state = oemf.emformer_infer(sd, cfg, chunk, torch.full((2,), chunk.shape[1]), state)
        feats.append(o[:, :emit])
    f = torch.cat(feats, 1)
    np.testing.assert_allclose(o1.cpu().numpy(), torch.nn.functional.linear(f, sd["proj1.weight"], sd["proj1.bias"]).numpy(), atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(o2.cpu().numpy(), torch.nn.functional.linear(f, sd["proj2.weight"], sd["proj2.bias"]).numpy(), atol=2e-4, rtol=1e-4)
    # single-head model: proj() on a slice of infer()'s output (not the identical tensor) still projects
    chp1 = configs.conan_hparams(True)
    m1 = EmformerDistillModel(chp1, output_dim=100)
    sd1_np = synth.emformer_state_dict(chp1, 0)
    m1.load_state_dict(_t(sd1_np), strict=True)
    sd1 = to_torch_sd(sd1_np)
    chunk = mel[:, :6]
    out, _, _ = m1.emformer.infer(chunk.cuda(), torch.full((2,), 6).cuda(), None)
    lg_full = m1.proj(out)
    lg_slice = m1.proj(out[:, :2])
    np.testing.assert_allclose(lg_slice.cpu().numpy(), lg_full[:, :2].cpu().numpy(), atol=2e-5, rtol=1e-5)
