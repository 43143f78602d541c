"""GPU tests of the other BASELINE.json configurations through size-independent properties and the oracle:
batch=64 full-size streams (slot independence), windowed mode (reset + ctx+chunk frames),
40 ms chunks (seg=2), and stream re-use after reset."""
import numpy as np
import pytest
import torch

from conan_amd import configs, synth
from tests.conftest import ARITHS, assert_arith_ran, kernels_of, load_golden

pytestmark = pytest.mark.gpu


def _full_ctx(chp=None):
    from conan_amd.runtime import Context
    chp = chp or configs.conan_hparams()
    vhp = configs.hifigan_hparams()
    ctx = Context(chp, vhp, 0)
    ctx.load_state_dict("emformer", synth.emformer_state_dict(chp, 0))
    ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0))
    ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
    ctx.finalize()
    return ctx, chp, vhp


@pytest.mark.parametrize("arith", ARITHS)
def test_batch64_full_size_streams_are_independent(arith):
    """configs[2]: 64 concurrent full-size streams.  A stream's output must not depend on which other streams share
    the batch or on its slot: stream k inside the batch of 64 == the same stream run alone (different tile shapes,
    split-K and launch grouping, so equality is up to fp32 re-association)."""
    from conan_amd.engine import StreamingVoiceConversionEngine
    ctx, chp, vhp = _full_ctx()
    B, T, Tr = 64, 12, 40
    src = torch.from_numpy(np.concatenate([synth.mel(T, 1234 + s) for s in range(B)])).cuda()
    ref = torch.from_numpy(np.concatenate([synth.mel(Tr, 4321 + s) for s in range(B)])).cuda()
    eng = StreamingVoiceConversionEngine(ctx, B, max_ref_frames=64, arith=arith)
    wav, mel, codes = eng.infer(src, ref)
    assert wav.shape == (B, T * 320) and mel.shape == (B, T, 80) and codes.shape == (B, T)
    assert torch.isfinite(wav).all() and float(wav.abs().max()) <= 1.0
    solo = StreamingVoiceConversionEngine(ctx, 1, max_ref_frames=64, arith=arith)
    for k in (0, 37, 63):
        w1, m1, c1 = solo.infer(src[k:k + 1], ref[k:k + 1])
        assert torch.equal(c1, codes[k:k + 1])
        np.testing.assert_allclose(m1.cpu().numpy(), mel[k:k + 1].cpu().numpy(), atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(w1.cpu().numpy(), wav[k:k + 1].cpu().numpy(), atol=2e-5, rtol=0)
    # re-running the same utterances after reset reproduces the result bit for bit (state fully re-initialised)
    wav2, mel2, _ = eng.infer(src, ref)
    assert torch.equal(wav, wav2) and torch.equal(mel, mel2)
    # ... and the pipelined schedule (default) is bit-identical to the blocking chunk loop
    wav3, mel3, codes3 = eng.infer(src, ref, pipelined=False)
    assert torch.equal(wav, wav3) and torch.equal(mel, mel3) and torch.equal(codes, codes3)
    names = kernels_of(eng.st, lambda: eng.st.hifigan_step(eng.slots, mel[:, :4].contiguous()))
    assert_arith_ran(names, arith)
    if arith == "limb":      # every ResBlock stage and ups.2 / ups.3 in limb form at 64 streams; no f32 ResBlock pass left
        assert all(any(("resblock_limb_kernel<%d," % c) in k for k in names) for c in (128, 64, 32)), sorted(names)
        assert not any("resblock_fused_kernel" in k or "resblock_pair_kernel" in k for k in names), sorted(names)
    eng.st.close(); solo.st.close(); ctx.close()


def test_large_stream_set_with_a_few_scattered_active_slots():
    """1024 slots (9 GB of per-stream state; the exchange and split-K workspaces must not grow with it), six of them active at
    scattered indices: the same streams in a 6-slot stream-set agree up to fp32 re-association / the arithmetic form of individual
    launches (include/conan_hip.h, conan_arith: what is and is not invariant), a re-run reproduces the bits, pipelined == blocking."""
    from conan_amd.engine import StreamingVoiceConversionEngine
    ctx, chp, vhp = _full_ctx()
    slots = [1000, 3, 517, 64, 999, 1]
    B, T, Tr = len(slots), 12, 40
    src = torch.from_numpy(np.concatenate([synth.mel(T, 77 + s) for s in range(B)])).cuda()
    ref = torch.from_numpy(np.concatenate([synth.mel(Tr, 177 + s) for s in range(B)])).cuda()
    free0 = torch.cuda.mem_get_info()[0]
    big = StreamingVoiceConversionEngine(ctx, 1024, max_ref_frames=64)
    used = free0 - torch.cuda.mem_get_info()[0]
    assert used < 16 * 2**30, f"{used / 2**30:.1f} GiB for 1024 slots"
    big.slots = slots
    wav, mel, codes = big.infer(src, ref)
    assert wav.shape == (B, T * 320) and torch.isfinite(wav).all()
    small = StreamingVoiceConversionEngine(ctx, B, max_ref_frames=64)
    w1, m1, c1 = small.infer(src, ref)
    assert torch.equal(c1, codes)
    np.testing.assert_allclose(m1.cpu().numpy(), mel.cpu().numpy(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(w1.cpu().numpy(), wav.cpu().numpy(), atol=2e-5, rtol=0)
    wav2, mel2, _ = big.infer(src, ref)
    assert torch.equal(wav, wav2) and torch.equal(mel, mel2)
    wav3, mel3, codes3 = big.infer(src, ref, pipelined=False)
    assert torch.equal(wav, wav3) and torch.equal(mel, mel3) and torch.equal(codes, codes3)
    big.st.close(); small.st.close(); ctx.close()


def test_windowed_mode_matches_reference_windows():
    """configs[1]/[4] windowed mode = state reset + (ctx + chunk) frames in one step, emit the last chunk.  Oracle:
    the reference module fed the same window (golden mel_out_win{8,16,32}, tools/make_goldens.py)."""
    ctx, chp, vhp = _full_ctx()
    g = load_golden("conan_full.npz")
    content = torch.from_numpy(g["content"]).int().cuda()
    T = content.shape[1]
    st = ctx.streams(1, max_frames=36, max_ref_frames=160)
    st.set_reference([0], torch.from_numpy(g["ref"]).cuda())
    for c in (8, 16, 32):
        st.reset([0], which=2)
        mel = st.decoder_step([0], content[:, T - c - 4:])
        np.testing.assert_allclose(mel[:, -4:].cpu().numpy(), g[f"mel_out_win{c}"], atol=1e-4, rtol=1e-4)
    # and the vocoder in windowed mode: reset + 12 frames == the first 12 frames of a stateful run
    gv = load_golden("hifigan_full.npz")
    mel12 = torch.from_numpy(gv["mel_12"]).transpose(1, 2).contiguous().cuda()
    st.reset([0], which=4)
    wav = st.hifigan_step([0], mel12)
    np.testing.assert_allclose(wav[0].cpu().numpy(), gv["wav_12"], atol=1e-4, rtol=0)
    st.close(); ctx.close()


def test_40ms_chunks_seg2():
    """configs[4]: chunk_size 40 -> Emformer segment 2 (+ right context 2), decoder / vocoder steps of 2 frames."""
    from oracle import emformer as oemf
    from oracle import loop as oloop
    from oracle.common import to_torch_sd
    chp = dict(configs.conan_hparams(), chunk_size=40)
    ctx, chp, vhp = _full_ctx(chp)
    assert ctx.cfg.emf_segment == 2
    sds = {"emformer": to_torch_sd(synth.emformer_state_dict(chp, 0)), "conan": to_torch_sd(synth.conan_state_dict(chp, 0)),
           "hifigan": to_torch_sd(synth.hifigan_state_dict(vhp, 0))}
    cfg = oemf.EmformerCfg(chp)
    B, T, Tr = 2, 11, 40
    src, ref = synth.mel(T, 1234, B), synth.mel(Tr, 4321, B)
    st = ctx.streams(B, max_frames=2, max_ref_frames=64)
    slots = [0, 1]
    st.reset(slots)
    st.set_reference(slots, torch.from_numpy(ref).cuda())
    wavs, mels = [], []
    for pos, emit, chunk in oemf.chunk_iter(torch.from_numpy(src), 2, 2):
        _, m, w = st.step(slots, chunk.cuda().contiguous(), emit=emit)
        wavs.append(w.cpu()); mels.append(m.cpu())
    wav, mel = torch.cat(wavs, 1), torch.cat(mels, 1)
    for b in range(B):
        w_ref, m_ref, _ = oloop.infer_once_stateful(sds["emformer"], cfg, sds["conan"], chp, sds["hifigan"], vhp, src[b], ref[b])
        np.testing.assert_allclose(mel[b].numpy(), m_ref, atol=1e-4, rtol=1e-4)
        np.testing.assert_allclose(wav[b].numpy(), w_ref, atol=1e-4, rtol=0)
    st.close(); ctx.close()


def test_right_context_zero_fast_system():
    """`right_context: 0` (the reference README's "fast system"): the Emformer sees segment-only
    chunks.  Fused step vs the oracle, 14 chunks so the left-context ring saturates and wraps."""
    from oracle import emformer as oemf
    from oracle.common import to_torch_sd
    chp = dict(configs.conan_hparams(), right_context=0)
    ctx, chp, vhp = _full_ctx(chp)
    assert ctx.cfg.emf_right_context == 0
    sd = to_torch_sd(synth.emformer_state_dict(chp, 0))
    cfg = oemf.EmformerCfg(chp)
    B, T = 3, 56
    mel = torch.from_numpy(synth.mel(T, 99, B))
    st = ctx.streams(B, max_frames=4, max_ref_frames=16)
    slots = [2, 0, 1]
    st.reset(slots)
    state = None
    for pos, emit, chunk in oemf.chunk_iter(mel, cfg.segment_length, 0):
        o_ref, _, state = oemf.emformer_infer(sd, cfg, chunk, torch.full((B,), chunk.shape[1]), state)
        lg_ref, codes_ref = oemf.logits_and_codes(sd, o_ref)
        o, lg, codes = st.emformer_step(slots, chunk.cuda())
        np.testing.assert_allclose(o.cpu().numpy(), o_ref.numpy(), atol=1e-4, rtol=1e-4)
        np.testing.assert_allclose(lg.cpu().numpy(), lg_ref.numpy(), atol=2e-4, rtol=1e-4)
        top2 = lg_ref.topk(2, -1).values
        safe = (top2[..., 0] - top2[..., 1]) > 1e-3
        assert torch.equal(codes.cpu().long()[safe], codes_ref[safe])
    st.close(); ctx.close()


@pytest.mark.parametrize("tag,vhp", [("zero_rb2_tiny", configs.HIFIGAN_ZERO_RB2_TINY), ("zero_rb2_full", configs.HIFIGAN_ZERO_RB2)])
def test_zero_insert_upsampler_and_resblock2(tag, vhp):
    """`upsample: zero` (polyphase form of CausalUpsampleBlock2) + `resblock: "2"` through the
    same conv kernel; streamed in 4-frame steps against the reference golden (one-shot forward) and the oracle taps."""
    from conan_amd.runtime import Context
    from oracle import hifigan as ohifi
    from oracle.common import to_torch_sd
    ctx = Context(None, vhp, 0, False, False, True)
    ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
    ctx.finalize()
    g = load_golden(f"hifigan_{tag}.npz")
    st = ctx.streams(2, max_frames=8, max_ref_frames=16)
    mel = torch.from_numpy(g["mel_40"]).transpose(1, 2).contiguous().cuda()        # [1, 40, 80]
    for slot, step in ((1, 4), (0, 8)):
        st.reset([slot])
        outs = [st.hifigan_step([slot], mel[:, i:i + step]) for i in range(0, 40, step)]
        wav = torch.cat(outs, 1)[0].cpu().numpy()
        np.testing.assert_allclose(wav, g["wav_40"], atol=1e-4, rtol=0)
    # pre-tanh against the oracle on the 12-frame case
    vsd = to_torch_sd(synth.hifigan_state_dict(vhp, 0))
    taps = {}
    ohifi.generator_forward(vsd, vhp, torch.from_numpy(g["mel_12"]), None, taps)
    st.reset([0])
    wav12, pre = st.hifigan_step([0], torch.from_numpy(g["mel_12"]).transpose(1, 2).contiguous().cuda()[:, :8], want_pre_tanh=True)
    ref_pre = taps["pre_tanh"][0, 0, :8 * 320].numpy()
    np.testing.assert_allclose(pre[0].cpu().numpy(), ref_pre, atol=1e-4 * max(1.0, np.abs(ref_pre).max()), rtol=0)
    st.close(); ctx.close()


@pytest.mark.parametrize("tag,vhp", [("nn_tiny", configs.HIFIGAN_NN_TINY), ("nn_full", configs.HIFIGAN_NN)])
def test_transposed_conv_upsampler_whole_forward(tag, vhp):
    """`upsample: nn` (CausalUpsampleBlock1, hifigan_causal.py:60-145).  The block looks two input
    frames ahead at every stage, so it runs as ONE step over the whole input from reset state (polyphase taps over rows
    t .. t+2, zeros beyond the end) - against the reference golden, per-stage taps included; a second step without a reset
    is refused; the Python seam's forward and a 2-slot batch give the same samples."""
    from conan_amd import _lib
    from conan_amd.runtime import Context
    from conan_amd.modules.vocoder.hifigan.hifigan_causal import HifiGanGenerator
    ctx = Context(None, vhp, 0, False, False, True)
    ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
    ctx.finalize()
    g = load_golden(f"hifigan_{tag}.npz")
    st = ctx.streams(2, max_frames=40, max_ref_frames=16)
    mel40 = torch.from_numpy(g["mel_40"]).transpose(1, 2).contiguous().cuda()      # [1, 40, 80]
    mel12 = torch.from_numpy(g["mel_12"]).transpose(1, 2).contiguous().cuda()
    wav = st.hifigan_step([1], mel40)                                                 # a new stream-set starts reset
    np.testing.assert_allclose(wav[0].cpu().numpy(), g["wav_40"], atol=1e-4, rtol=0)
    with pytest.raises(_lib.ConanError, match="looks ahead"):
        st.hifigan_step([1], mel40[:, :4])
    st.reset([1, 0])
    both = st.hifigan_step([1, 0], torch.cat([mel40, mel40.flip(1)], 0))
    np.testing.assert_allclose(both[0].cpu().numpy(), g["wav_40"], atol=1e-4, rtol=0)
    st.reset([0])
    wav12, _, _, ups = st.hifigan_step_taps([0], mel12)
    for i in range(len(vhp["upsample_rates"])):
        ref = g[f"ups.{i}_12"].T                                                      # [T_i, C_i]
        np.testing.assert_allclose(ups[i][0].cpu().numpy(), ref, atol=1e-4 * max(1.0, np.abs(ref).max()), rtol=0)
    np.testing.assert_allclose(wav12[0].cpu().numpy(), g["wav_12"], atol=1e-4, rtol=0)
    st.close(); ctx.close()
    gen = HifiGanGenerator(dict(vhp))
    gen.load_state_dict({k: torch.from_numpy(v) for k, v in synth.hifigan_state_dict(vhp, 0).items()}, strict=True)
    y = gen(torch.from_numpy(g["mel_40"]).cuda())
    np.testing.assert_allclose(y[0, 0].cpu().numpy(), g["wav_40"], atol=1e-4, rtol=0)
    y = gen(torch.from_numpy(g["mel_12"]).cuda())                                    # resets between forwards
    np.testing.assert_allclose(y[0, 0].cpu().numpy(), g["wav_12"], atol=1e-4, rtol=0)
    if tag == "nn_tiny":       # a 3 s utterance (rings regrown to 150 frames = 48 000 rows in the last stage) against the oracle
        from oracle import hifigan as ohifi
        from oracle.common import to_torch_sd
        mel = torch.from_numpy(synth.mel(150, 77, 2)).transpose(1, 2).contiguous()   # [2, 80, 150]
        ref = ohifi.generator_forward(to_torch_sd(synth.hifigan_state_dict(vhp, 0)), vhp, mel)[:, 0].numpy()
        y = gen(mel.cuda())
        np.testing.assert_allclose(y[:, 0].cpu().numpy(), ref, atol=1e-4, rtol=0)


def test_emformer_mode_both_uses_proj1():
    """`mode: both` (modules/Emformer/emformer.py:28-30): checkpoints carry proj1 (80 -> 100) and proj2 (80 -> 768); the
    streaming loop projects with proj1 (inference/Conan.py:117-118).  Codes / logits against the oracle, and through the
    reference-shaped module (`emformer.proj1(out)`)."""
    from conan_amd.modules.Emformer.emformer import EmformerDistillModel
    from oracle import emformer as oemf
    from oracle.common import to_torch_sd
    chp = dict(configs.conan_hparams(), mode="both", emformer_output_dim=768)
    sd_np = synth.emformer_state_dict(chp, 0, output_dim=768)
    assert "proj1.weight" in sd_np and sd_np["proj2.weight"].shape == (768, 80)
    sd = to_torch_sd(sd_np)
    cfg = oemf.EmformerCfg(chp)
    model = EmformerDistillModel(chp, output_dim=768)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    B, T = 2, 24
    mel = torch.from_numpy(synth.mel(T, 5, B))
    state_ref, state = None, None
    for pos, emit, chunk in oemf.chunk_iter(mel, cfg.segment_length, cfg.right_context_length):
        lengths = torch.full((B,), chunk.shape[1])
        o_ref, _, state_ref = oemf.emformer_infer(sd, cfg, chunk, lengths, state_ref)
        lg_ref, codes_ref = oemf.logits_and_codes(sd, o_ref)
        assert lg_ref.shape[-1] == 100
        out, _, state = model.emformer.infer(chunk.cuda(), lengths.cuda(), state)
        lg = model.proj1(out)
        np.testing.assert_allclose(lg.cpu().numpy(), lg_ref.numpy(), atol=2e-4, rtol=1e-4)
        top2 = lg_ref.topk(2, -1).values
        safe = (top2[..., 0] - top2[..., 1]) > 1e-3
        assert torch.equal(lg.argmax(-1).cpu()[safe], codes_ref[safe])
        lg2 = model.proj2(out)                       # the second head stays a plain Linear
        np.testing.assert_allclose(lg2.cpu().numpy(), torch.nn.functional.linear(o_ref, sd["proj2.weight"], sd["proj2.bias"]).numpy(), atol=2e-4, rtol=1e-4)
