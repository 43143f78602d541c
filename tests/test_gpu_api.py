"""GPU tests through the reference-shaped Python interfaces (module/forward API, vocoder registry, inference
driver) -- the seams SURVEY.md §8b lists -- against the golden vectors and the CPU oracle."""
import numpy as np
import pytest
import torch

from conan_amd import configs, synth
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


def _t(sd):
    return {k: torch.from_numpy(v) for k, v in sd.items()}


def test_generator_forward_and_spec2wav_match_reference_golden():
    from conan_amd.modules.vocoder.hifigan.hifigan_causal import HifiGanGenerator
    from conan_amd.tasks.tts.vocoder_infer.base_vocoder import get_vocoder_cls
    import conan_amd.tasks.tts.vocoder_infer.hifigan  # noqa: F401  (registers 'HifiGAN')
    vhp = configs.hifigan_hparams()
    g = load_golden("hifigan_full.npz")
    gen = HifiGanGenerator(vhp)
    gen.load_state_dict(_t(synth.hifigan_state_dict(vhp, 0)))
    y = gen(torch.from_numpy(g["mel_150"]).cuda())
    assert y.shape == (1, 1, 150 * 320)
    np.testing.assert_allclose(y[0, 0].cpu().numpy(), g["wav_150"], atol=1e-4, rtol=0)
    voc = get_vocoder_cls("HifiGAN")(config=vhp, state_dict=_t(synth.hifigan_state_dict(vhp, 0)))
    wav = voc.spec2wav(g["mel_12"][0].T)                      # numpy [T,80] -> numpy [T*hop]
    assert wav.dtype == np.float32 and wav.shape == (12 * 320,)
    np.testing.assert_allclose(wav, g["wav_12"], atol=1e-4, rtol=0)
    assert get_vocoder_cls("NoSuchVocoder") is None


def test_conan_forward_matches_reference_golden():
    from conan_amd.modules.Conan.Conan import Conan
    chp = configs.conan_hparams()
    g = load_golden("conan_full.npz")
    m = Conan(0, chp)
    m.load_state_dict(_t(synth.conan_state_dict(chp, 0)), strict=True)
    ret = m(content=torch.from_numpy(g["content"]).cuda(), ref=torch.from_numpy(g["ref"]).cuda(), infer=True, global_steps=200000)
    assert ret["mel_out"].shape == (1, 150, 80)
    assert np.array_equal(ret["pitch_bins"].cpu().numpy(), g["pitch_bins"])
    np.testing.assert_allclose(ret["uv_pred"].cpu().numpy(), g["uv_pred"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(ret["decoder_inp"].cpu().numpy(), g["decoder_inp"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(ret["mel_out"].cpu().numpy(), g["mel_out"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(ret["f0_denorm_pred"].cpu().numpy(), g["f0_denorm_pred"], atol=2e-2, rtol=1e-4)
    # the remaining dict entries of SURVEY.md §8b seam (3)
    assert ret["style_embed"].shape == (1, 1, 256) and ret["content_embed_proj"].shape == (1, 150, 256)
    np.testing.assert_allclose(ret["style_embed"].cpu().numpy(), g["style_embed"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(ret["content_embed_proj"].cpu().numpy(), g["content_embed_proj"], atol=1e-5, rtol=1e-5)
    assert len(ret["attn"]) == 2 and ret["attn"][0].shape == (1, 1, 150, 38)
    np.testing.assert_allclose(ret["attn"][0].cpu().numpy(), g["attn0"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(ret["attn"][1].sum(-1).cpu().numpy(), 1.0, atol=1e-5)


def test_emformer_module_streaming_api():
    from conan_amd.modules.Emformer.emformer import EmformerDistillModel
    from oracle import emformer as oemf
    from oracle.common import to_torch_sd
    chp = configs.conan_hparams()
    sd = synth.emformer_state_dict(chp, 0)
    e = EmformerDistillModel(chp, output_dim=100)
    e.load_state_dict(_t(sd), strict=True)
    mel = torch.from_numpy(synth.mel(30, 1234, 2))
    logits = e.inference(mel.cuda())
    ref_logits, _ = oemf.stream_codes(to_torch_sd(sd), oemf.EmformerCfg(chp), mel)
    np.testing.assert_allclose(logits.cpu().numpy(), ref_logits.numpy(), atol=2e-4, rtol=1e-4)
    with pytest.raises(ValueError):        # torchaudio raises ValueError for a wrong chunk length
        e.emformer.infer(mel[:, :5].cuda(), torch.tensor([5, 5]).cuda(), None)
    # inference_rtf (modules/Emformer/emformer.py:99-156): the same logits + one latency and one real-time factor per chunk step
    out, lat, rtf = e.inference_rtf(mel.cuda())
    assert torch.equal(out, logits)
    n_steps = -(-30 // e.segment_length)
    assert len(lat) == n_steps and len(rtf) == n_steps and all(0 < x < 1.0 for x in lat)
    assert all(abs(r - l / (e.segment_length * 0.02)) < 1e-12 for r, l in zip(rtf, lat))


def test_streaming_voice_conversion_infer_once():
    from conan_amd.inference.Conan import StreamingVoiceConversion
    from oracle import emformer as oemf
    from oracle import loop as oloop
    from oracle.common import to_torch_sd
    chp, vhp = configs.conan_hparams(True), configs.hifigan_hparams(True)
    sds = {"emformer": synth.emformer_state_dict(chp, 0), "conan": synth.conan_state_dict(chp, 0), "hifigan": synth.hifigan_state_dict(vhp, 0)}
    eng = StreamingVoiceConversion(chp, vhp, sds)
    src, ref = synth.mel(31, 1234)[0], synth.mel(40, 4321)[0]
    wav, mel = eng.infer_once({"src_mel": src, "ref_mel": ref})
    t = {k: to_torch_sd(v) for k, v in sds.items()}
    w_ref, m_ref, _ = oloop.infer_once_ref(t["emformer"], oemf.EmformerCfg(chp), t["conan"], chp, t["hifigan"], vhp, src, ref)
    assert wav.shape == (31 * 320,) and mel.shape == (31, 80)
    np.testing.assert_allclose(mel, m_ref, atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(wav, w_ref, atol=1e-4, rtol=0)
    with pytest.raises(ValueError):
        StreamingVoiceConversion(dict(chp, vocoder="Nope"), vhp, sds)


def test_streaming_voice_conversion_with_lookahead_vocoder():
    """A vocoder config whose upsampler looks ahead (`upsample: nn`) through StreamingVoiceConversion: like the reference
    loop (inference/Conan.py:147-155) the vocoder runs on all mel frames so far and the current chunk's samples are kept,
    so each chunk's tail lacks its look-ahead exactly as in the reference - against the reference-semantics oracle loop."""
    from conan_amd.inference.Conan import StreamingVoiceConversion
    from oracle import emformer as oemf
    from oracle import loop as oloop
    from oracle.common import to_torch_sd
    chp, vhp = configs.conan_hparams(True), dict(configs.HIFIGAN_NN_TINY)
    sds = {"emformer": synth.emformer_state_dict(chp, 0), "conan": synth.conan_state_dict(chp, 0), "hifigan": synth.hifigan_state_dict(vhp, 0)}
    eng = StreamingVoiceConversion(chp, vhp, sds)
    src, ref = synth.mel(23, 1234)[0], synth.mel(40, 4321)[0]
    wav, mel = eng.infer_once({"src_mel": src, "ref_mel": ref})
    t = {k: to_torch_sd(v) for k, v in sds.items()}
    w_ref, m_ref, _ = oloop.infer_once_ref(t["emformer"], oemf.EmformerCfg(chp), t["conan"], chp, t["hifigan"], vhp, src, ref)
    assert wav.shape == (23 * 320,) and mel.shape == (23, 80)
    np.testing.assert_allclose(mel, m_ref, atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(wav, w_ref, atol=1e-4, rtol=0)
    wav2, _ = eng.infer_once({"src_mel": src[:9], "ref_mel": ref})        # a second, shorter utterance on the same engine
    np.testing.assert_allclose(wav2[:4 * 320], wav[:4 * 320], atol=1e-6, rtol=0)
    # the fused chunk steps carry vocoder state from chunk to chunk: refused for this vocoder before anything is enqueued
    from conan_amd import _lib
    st = eng.engine.st
    chunk = torch.from_numpy(src[None, :6]).cuda()
    for call in (lambda: st.step([0], chunk), lambda: st.step_async([0], chunk, torch.empty(1, 4 * 320, device="cuda"))):
        with pytest.raises(_lib.ConanError) as ei:
            call()
        assert ei.value.code == _lib.ERR_UNSUPPORTED


def test_error_conventions_through_the_c_abi():
    from conan_amd import _lib
    from conan_amd.runtime import Context
    chp, vhp = configs.conan_hparams(True), configs.hifigan_hparams(True)
    ctx = Context(chp, vhp, 0)
    with pytest.raises(_lib.ConanError) as ei:      # finalize with nothing loaded names the first missing tensor
        ctx.finalize()
    assert ei.value.code == _lib.ERR_MISSING and "conv_pre" in str(ei.value)
    ctx.close()
    ctx = Context(chp, vhp, 0)
    for k, sd in (("emformer", synth.emformer_state_dict(chp, 0)), ("conan", synth.conan_state_dict(chp, 0)), ("hifigan", synth.hifigan_state_dict(vhp, 0))):
        ctx.load_state_dict(k, sd)
    ctx.finalize()
    st = ctx.streams(2, 4, 16)
    with pytest.raises(_lib.ConanError) as ei:      # decoder before set_reference (reference: ValueError when ref is None)
        st.decoder_step([0], torch.zeros(1, 4, dtype=torch.int32).cuda())
    assert ei.value.code == _lib.ERR_STATE
    with pytest.raises(_lib.ConanError):            # slot out of range
        st.reset([5])
    with pytest.raises(_lib.ConanError):            # more frames than the streams were created for
        st.hifigan_step([0], torch.zeros(1, 9, 80).cuda())
    st.close()
    ctx.close()


def test_pipelined_step_matches_fused_step():
    """conan_step_async (front-end of chunk t+1 overlapping the vocoder of chunk t on internal HIP streams) against
    conan_step on a second stream-set: bit-identical codes, mel and audio over 12 chunks, mixed with a reset."""
    import numpy as np
    from conan_amd import configs, synth
    from conan_amd.runtime import Context
    chp, vhp = configs.conan_hparams(True), configs.hifigan_hparams(True)
    ctx = Context(chp, vhp, 0, True, True, True)
    ctx.load_state_dict("emformer", synth.emformer_state_dict(chp, 0))
    ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0))
    ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
    ctx.finalize()
    B, T = 5, 12 * 4 + 2
    a, b = ctx.streams(B, 4, 64), ctx.streams(B, 4, 64)
    slots = [4, 0, 2, 1, 3]
    ref = torch.from_numpy(synth.mel(40, 3, B)).cuda()
    src = torch.from_numpy(synth.mel(T, 5, B)).cuda()
    for st in (a, b):
        st.reset(slots)
        st.set_reference(slots, ref)
    hop = ctx.hop
    outs_a, outs_b = [], []
    for t in range(12):
        chunk = src[:, 4 * t:4 * t + 6].contiguous()
        codes, mel, wav = a.step(slots, chunk)
        outs_a.append((codes.clone(), mel.clone(), wav.clone()))
        cb = torch.empty(B, 4, dtype=torch.int32, device="cuda"); mb = torch.empty(B, 4, 80, device="cuda"); wb = torch.empty(B, 4 * hop, device="cuda")
        b.step_async(slots, chunk, wb, codes=cb, mel_out=mb)
        outs_b.append((cb, mb, wb))
    b.join()
    torch.cuda.synchronize()
    for (ca, ma, wa), (cb, mb, wb) in zip(outs_a, outs_b):
        assert torch.equal(ca, cb) and torch.equal(ma, mb) and torch.equal(wa, wb)
    # a blocking call after pipelined ones joins them first
    b.step_async(slots, src[:, :6].contiguous(), torch.empty(B, 4 * hop, device="cuda"))
    b.reset(slots)
    a.reset(slots)
    chunk = src[:, :6].contiguous()
    _, _, w1 = a.step(slots, chunk)
    _, _, w2 = b.step(slots, chunk)
    assert torch.equal(w1, w2)
    a.close(); b.close(); ctx.close()


def test_wav2mel_frontend_matches_oracle():
    """conan_wav2mel (frames -> f64 DFT sums -> magnitude -> f64 filterbank sums -> log10 / clip, all on the GPU) against
    the numpy restatement of librosa_wav2spec (oracle/frontend.py, itself pinned against transformers.audio_utils and
    torch.stft in tests/test_oracle_frontend.py).  Tolerance 1e-5 in log10 units everywhere, including a pure tone whose
    quiet mel bins lie 6 decades below the peak (measured: <= 1e-6)."""
    from conan_amd import configs
    from conan_amd.runtime import Context
    from oracle import frontend as ofe
    chp = configs.conan_hparams(True)
    ctx = Context(chp, None, 0, False, True, False)
    ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0))
    ctx.finalize()
    rng = np.random.default_rng(3)
    sr, n = 16000, 3
    t = np.arange(int(1.3 * sr) + 57) / sr
    wavs = np.stack([(0.5 * np.sin(2 * np.pi * (220 * (i + 1)) * t) * np.exp(-t) + 0.1 * np.sin(2 * np.pi * 3100 * t)
                      + 0.02 * rng.standard_normal(t.shape)).astype(np.float32) for i in range(n)])
    mel = ctx.wav2mel(torch.from_numpy(wavs).cuda()).cpu().numpy()
    for i in range(n):
        ref = ofe.wav2mel(wavs[i])
        assert mel[i].shape == ref.shape == (1 + wavs.shape[1] // 320, 80)
        assert np.abs(mel[i] - ref).max() < 1e-5, np.abs(mel[i] - ref).max()
    tone = (0.9 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    mt, rt = ctx.wav2mel(torch.from_numpy(tone[None]).cuda()).cpu().numpy()[0], ofe.wav2mel(tone)
    assert rt.min() == -6.0 and rt.max() > 0.9 and np.abs(mt - rt).max() < 1e-5, np.abs(mt - rt).max()
    # silence -> floor; a second call with another length reuses the tables
    z = ctx.wav2mel(torch.zeros(1, 4000, device="cuda")).cpu().numpy()
    assert z.shape == (1, 13, 80) and np.all(z == -6.0)
    # the earlier loop's front-end (inference/Conan_previous.py:100-121: reflect padding, torch.stft(center=False),
    # sqrt(. + 1e-9), natural log with floor 1e-5) through the same kernels; ln magnifies by ln(10): 3e-5
    from conan_amd.utils.audio import mel_spectrogram
    got = mel_spectrogram(torch.from_numpy(wavs), 1024, 80, sr, 320, 1024, 80, None, center=False, ctx=ctx).cpu().numpy()
    for i in range(n):
        ref = ofe.torch_mel_spectrogram(wavs[i])
        assert got[i].shape == ref.shape == (80, wavs.shape[1] // 320)
        assert np.abs(got[i] - ref).max() < 3e-5, np.abs(got[i] - ref).max()
    gt = mel_spectrogram(torch.from_numpy(tone[None]), 1024, 80, sr, 320, 1024, 80, None, ctx=ctx).cpu().numpy()[0]
    rt2 = ofe.torch_mel_spectrogram(tone)
    assert np.abs(rt2.min() - np.log(1e-5)) < 1e-6 and np.abs(gt - rt2).max() < 3e-5, np.abs(gt - rt2).max()
    ctx.close()


def test_streaming_vc_accepts_waveforms(tmp_path):
    """StreamingVoiceConversion.infer_once({'ref_wav','src_wav'}) (inference/Conan.py:72-80): wav path / array ->
    GPU mel front-end -> chunk loop; equals the mel-input call on the oracle's mels of the same waveforms."""
    import wave
    from conan_amd.inference.Conan import StreamingVoiceConversion
    from oracle import frontend as ofe
    chp, vhp = configs.conan_hparams(True), configs.hifigan_hparams(True)
    sds = {"emformer": _t(synth.emformer_state_dict(chp, 0)), "conan": _t(synth.conan_state_dict(chp, 0)),
           "hifigan": _t(synth.hifigan_state_dict(vhp, 0))}
    vc = StreamingVoiceConversion(chp, vhp, sds)
    sr = 16000
    t = np.arange(int(0.5 * sr)) / sr
    src = (0.4 * np.sin(2 * np.pi * 330 * t) + 0.1 * np.sin(2 * np.pi * 1900 * t)).astype(np.float32)
    ref = (0.3 * np.sin(2 * np.pi * 180 * t) * np.cos(2 * np.pi * 3 * t)).astype(np.float32)
    path = str(tmp_path / "src.wav")
    pcm = np.round(src * 32767).astype("<i2")
    with wave.open(path, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(sr); f.writeframes(pcm.tobytes())
    wav_a, mel_a = vc.infer_once({"ref_wav": ref, "src_wav": path})
    T = 1 + len(src) // 320
    assert mel_a.shape == (T, 80) and wav_a.shape == (T * 320,) and np.isfinite(wav_a).all()
    src_q = pcm.astype(np.float32) / 32768.0
    m_src, m_ref = vc._wav_to_mel(src_q).cpu().numpy(), vc._wav_to_mel(ref).cpu().numpy()
    assert np.abs(m_src - ofe.wav2mel(src_q)).max() < 1e-5 and np.abs(m_ref - ofe.wav2mel(ref)).max() < 1e-5
    wav_b, mel_b = vc.infer_once({"ref_mel": m_ref, "src_mel": m_src})
    assert np.array_equal(wav_a, wav_b) and np.array_equal(mel_a, mel_b)
    with pytest.raises(ValueError):
        vc.infer_once({"src_wav": src})


def test_batch_file_runner_matches_single_conversions(tmp_path):
    """VoiceConversionRunner (inference/run_voice_conversion.py): three pairs of different lengths converted as one
    batch of streams == the same pairs converted one by one through infer_once (wav files compared sample for sample)."""
    import json
    from scipy.io import wavfile
    from conan_amd.inference.run_voice_conversion import VoiceConversionRunner
    from conan_amd.utils.audio.io import save_wav
    chp, vhp = configs.conan_hparams(True), configs.hifigan_hparams(True)
    sds = {"emformer": _t(synth.emformer_state_dict(chp, 0)), "conan": _t(synth.conan_state_dict(chp, 0)),
           "hifigan": _t(synth.hifigan_state_dict(vhp, 0))}
    sr = 16000
    rng = np.random.default_rng(5)
    pairs = []
    for k, (ds, dr) in enumerate(((0.50, 0.40), (0.33, 0.61), (0.42, 0.30))):
        ts, tr = np.arange(int(ds * sr)) / sr, np.arange(int(dr * sr)) / sr
        s = 0.4 * np.sin(2 * np.pi * (200 + 90 * k) * ts) + 0.02 * rng.standard_normal(ts.shape)
        r = 0.3 * np.sin(2 * np.pi * (150 + 40 * k) * tr) * np.cos(2 * np.pi * 2 * tr)
        save_wav(s, str(tmp_path / f"s{k}.wav"), sr); save_wav(r, str(tmp_path / f"r{k}.wav"), sr)
        pairs.append({"src_wav": str(tmp_path / f"s{k}.wav"), "ref_wav": str(tmp_path / f"r{k}.wav"), "output_name": f"out{k}.wav"})
    cfg = tmp_path / "pairs.json"
    cfg.write_text(json.dumps({"total_pairs": len(pairs), "conversion_pairs": pairs}))
    runner = VoiceConversionRunner(str(cfg), chp, vhp, sds, output_dir=str(tmp_path / "batched"), streams=4)
    res = runner.run_all_conversions()
    assert res["successful"] == 3 and res["failed"] == 0
    single = VoiceConversionRunner(str(cfg), chp, vhp, sds, output_dir=str(tmp_path / "single"), streams=1)
    for k, p in enumerate(pairs):
        ok, path = single.run_single_conversion(p, k)
        assert ok, path
        a = wavfile.read(str(tmp_path / "batched" / f"out{k}.wav"))[1]
        b = wavfile.read(path)[1]
        assert a.shape == b.shape and a.dtype == np.int16
        # different batch sizes pick different tile shapes / split-K: equal up to fp32 re-association = +-1 LSB of int16
        assert np.abs(a.astype(np.int32) - b.astype(np.int32)).max() <= 1


def test_pipelined_steps_with_changing_slot_lists_and_short_emit():
    """Pipelined steps whose slot list (and batch size) changes from call to call, with emit < segment on the way: the
    library drains the in-flight vocoder before it rewrites the slot table.  Reference: the same call sequence with
    blocking conan_step on a second stream-set; results must be bit-identical."""
    from conan_amd.runtime import Context
    chp, vhp = configs.conan_hparams(True), configs.hifigan_hparams(True)
    ctx = Context(chp, vhp, 0, True, True, True)
    ctx.load_state_dict("emformer", synth.emformer_state_dict(chp, 0))
    ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0))
    ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
    ctx.finalize()
    S = 4
    a, b = ctx.streams(S, 4, 64), ctx.streams(S, 4, 64)
    ref = torch.from_numpy(synth.mel(36, 8, S)).cuda()
    src = torch.from_numpy(synth.mel(64, 9, S)).cuda()
    for st in (a, b):
        st.reset(list(range(S)))
        st.set_reference(list(range(S)), ref)
    hop = ctx.hop
    pos = [0] * S                                   # per-stream frame cursor
    plan = [([0, 1, 2, 3], 4), ([2, 0], 4), ([1, 3], 4), ([3], 2), ([0, 1, 2, 3], 4), ([1], 4), ([2, 3, 0], 3), ([0, 1, 2, 3], 4)]
    outs_a, outs_b = [], []
    for slots, emit in plan:
        chunk = torch.stack([src[s, pos[s]:pos[s] + 6] for s in slots]).contiguous()
        if emit < 4:    # a short final-style chunk: repeat-last padding of the look-ahead (inference/Conan.py:100-110)
            chunk = torch.cat([chunk[:, :emit], chunk[:, emit - 1:emit].expand(-1, 6 - emit, -1)], 1).contiguous()
        n = len(slots)
        c, m, w = a.step(slots, chunk, emit=emit)
        outs_a.append((c.clone(), m.clone(), w.clone()))
        cb = torch.empty(n, 4, dtype=torch.int32, device="cuda"); mb = torch.empty(n, emit, 80, device="cuda"); wb = torch.empty(n, emit * hop, device="cuda")
        b.step_async(slots, chunk, wb, emit=emit, codes=cb, mel_out=mb)
        outs_b.append((cb, mb, wb))
        for s in slots:
            pos[s] += emit
    b.join()
    torch.cuda.synchronize()
    for k, ((ca, ma, wa), (cb, mb, wb)) in enumerate(zip(outs_a, outs_b)):
        assert torch.equal(ca, cb) and torch.equal(ma, mb) and torch.equal(wa, wb), f"step {k}"
    a.close(); b.close(); ctx.close()
