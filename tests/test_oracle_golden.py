"""Pin the CPU oracle against golden vectors produced by the REFERENCE itself
(tools/make_goldens.py, run in the build container) and against the reference's own
invariants (hifigan_causal.py:550-680: causality, prefix consistency)."""
import numpy as np
import pytest
import torch

from conan_amd import configs, synth
from oracle import conan as oconan
from oracle import hifigan as ohifi
from oracle import loop as oloop
from oracle.common import to_torch_sd

TOL = 5e-6  # CPU restatement vs import-oracle


@pytest.fixture(scope="module", params=["tiny", "full"])
def models(request):
    tiny = request.param == "tiny"
    chp, vhp = configs.conan_hparams(tiny), configs.hifigan_hparams(tiny)
    return (request.param, chp, vhp, to_torch_sd(synth.conan_state_dict(chp, 0)),
            to_torch_sd(synth.hifigan_state_dict(vhp, 0)))


def test_hifigan_matches_reference(models, golden):
    tag, _, vhp, _, vsd = models
    g = golden(f"hifigan_{tag}.npz")
    for T in (12, 150):
        taps = {}
        wav = ohifi.generator_forward(vsd, vhp, torch.from_numpy(g[f"mel_{T}"]), None, taps)
        assert wav.shape == (1, 1, T * 320)
        np.testing.assert_allclose(wav[0, 0].numpy(), g[f"wav_{T}"], atol=TOL, rtol=0)
        if T == 12:
            np.testing.assert_allclose(taps["conv_pre"][0].numpy(), g["conv_pre_12"], atol=TOL, rtol=1e-5)
            np.testing.assert_allclose(taps["pre_tanh"][0].numpy(), g["pre_tanh_12"], atol=2e-5, rtol=1e-5)
            for i in range(4):
                np.testing.assert_allclose(taps[f"ups.{i}"][0].numpy(), g[f"ups.{i}_12"], atol=2e-5, rtol=1e-5)


def test_hifigan_stateful_equals_prefix(models, golden):
    """Stateful 4-frame steps == one-shot forward (the property the reference loop relies on)."""
    tag, _, vhp, _, vsd = models
    g = golden(f"hifigan_{tag}.npz")
    mel = torch.from_numpy(g["mel_12"])
    st, outs = {}, []
    for i in range(0, 12, 4):
        outs.append(ohifi.generator_forward(vsd, vhp, mel[:, :, i:i + 4], st))
    np.testing.assert_allclose(torch.cat(outs, 2)[0, 0].numpy(), g["wav_12"], atol=TOL, rtol=0)


def test_hifigan_reference_invariants(models):
    """verify_causality / verify_prefix_consistency of hifigan_causal.py:550-680 on the oracle."""
    _, _, vhp, _, vsd = models
    torch.manual_seed(0)
    x8 = torch.randn(1, 80, 8)
    x16 = torch.cat([x8, torch.randn(1, 80, 8)], 2)
    y8 = ohifi.generator_forward(vsd, vhp, x8)
    y16 = ohifi.generator_forward(vsd, vhp, x16)
    assert torch.allclose(y8, y16[:, :, :8 * 320], atol=1e-6)
    xp = x16.clone()
    xp[:, :, 9:] += torch.randn(1, 80, 7)
    yp = ohifi.generator_forward(vsd, vhp, xp)
    assert torch.allclose(yp[:, :, :9 * 320], y16[:, :, :9 * 320], atol=1e-6)
    assert not torch.allclose(yp[:, :, 9 * 320:], y16[:, :, 9 * 320:], atol=1e-6)


def test_conan_matches_reference(models, golden):
    tag, chp, _, csd, _ = models
    g = golden(f"conan_{tag}.npz")
    ret = oconan.conan_forward(csd, chp, torch.from_numpy(g["content"]), torch.from_numpy(g["ref"]))
    # integer intermediates: exact
    np.testing.assert_array_equal(ret["vq_ids"].numpy(), g["vq_ids"])
    np.testing.assert_array_equal(ret["pitch_bins"].numpy(), g["pitch_bins"])
    for k in ("style_embed", "content_embed_proj", "pitch_embed", "uv_pred", "decoder_inp", "mel_out"):
        np.testing.assert_allclose(ret[k].numpy(), g[k], atol=TOL, rtol=1e-5, err_msg=k)
    np.testing.assert_allclose(ret["f0_denorm_pred"].numpy(), g["f0_denorm_pred"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(ret["attn"][0].numpy(), g["attn0"], atol=TOL)
    # the reference's output dict keys
    for k in g["keys"]:
        assert str(k) in ret, k


def test_conan_stateful_and_windowed(models, golden):
    tag, chp, _, csd, _ = models
    g = golden(f"conan_{tag}.npz")
    content, ref = torch.from_numpy(g["content"]), torch.from_numpy(g["ref"])
    cache = oconan.style_pass(csd, chp, ref)
    st, mels = {}, []
    for i in range(0, 40, 4):
        mels.append(oconan.decode_frames(csd, chp, content[:, i:i + 4], cache, st)["mel_out"])
    np.testing.assert_allclose(torch.cat(mels, 1).numpy(), g["mel_out"][:, :40], atol=TOL, rtol=1e-5)
    T = content.shape[1]
    for c in (8, 16, 32):
        r = oconan.decode_frames(csd, chp, content[:, T - c - 4:], cache, None)["mel_out"][:, -4:]
        np.testing.assert_allclose(r.numpy(), g[f"mel_out_win{c}"], atol=TOL, rtol=1e-5)


def test_loop_matches_reference(models, golden):
    """inference/Conan.py:95-156 semantics with given codes: ref-semantics == stateful == golden."""
    tag, chp, vhp, csd, vsd = models
    from oracle import emformer as oemf
    g = golden(f"loop_{tag}.npz")
    ehp = configs.conan_hparams(tag == "tiny")
    esd = to_torch_sd(synth.emformer_state_dict(ehp, 0))
    cfg = oemf.EmformerCfg(ehp)
    T = len(g["codes"])
    src = synth.mel(T, 1234)[0]
    for fn in (oloop.infer_once_ref, oloop.infer_once_stateful):
        wav, mel, codes = fn(esd, cfg, csd, chp, vsd, vhp, src, g["ref"][0], codes_override=g["codes"])
        np.testing.assert_array_equal(codes, g["codes"])
        np.testing.assert_allclose(mel, g["mel"], atol=TOL, rtol=1e-5)
        np.testing.assert_allclose(wav, g["wav"], atol=2e-5, rtol=0)


@pytest.mark.parametrize("tag,vhp", [("zero_rb2_tiny", configs.HIFIGAN_ZERO_RB2_TINY), ("zero_rb2_full", configs.HIFIGAN_ZERO_RB2)])
def test_hifigan_zero_upsampler_resblock2_matches_reference(tag, vhp, golden):
    """The generator's other config.yaml choices (`upsample: zero`, `resblock: "2"`; hifigan_causal.py:151-165,
    :246-267, :287-303) against the reference generator built from that config, plus stateful == one-shot."""
    vsd = to_torch_sd(synth.hifigan_state_dict(vhp, 0))
    g = golden(f"hifigan_{tag}.npz")
    for T in (12, 40):
        taps = {}
        wav = ohifi.generator_forward(vsd, vhp, torch.from_numpy(g[f"mel_{T}"]), None, taps)
        np.testing.assert_allclose(wav[0, 0].numpy(), g[f"wav_{T}"], atol=TOL, rtol=0)
        if T == 12:
            for i in range(4):
                np.testing.assert_allclose(taps[f"ups.{i}"][0].numpy(), g[f"ups.{i}_12"], atol=2e-5, rtol=1e-5)
            np.testing.assert_allclose(taps["pre_tanh"][0].numpy(), g["pre_tanh_12"], atol=2e-5, rtol=1e-5)
    mel = torch.from_numpy(g["mel_40"])
    st, outs = {}, []
    for i in range(0, 40, 4):
        outs.append(ohifi.generator_forward(vsd, vhp, mel[:, :, i:i + 4], st))
    np.testing.assert_allclose(torch.cat(outs, 2)[0, 0].numpy(), g["wav_40"], atol=TOL, rtol=0)


@pytest.mark.parametrize("tag,vhp", [("nn_tiny", configs.HIFIGAN_NN_TINY), ("nn_full", configs.HIFIGAN_NN)])
def test_hifigan_transposed_conv_upsampler_matches_reference(tag, vhp, golden):
    """`upsample: nn` (CausalUpsampleBlock1, hifigan_causal.py:60-145) against the reference generator built from that
    config; and the property that makes it a whole-utterance-only block: it looks AHEAD (a later frame changes earlier
    samples), so prefix outputs differ from the full forward near the prefix end."""
    vsd = to_torch_sd(synth.hifigan_state_dict(vhp, 0))
    g = golden(f"hifigan_{tag}.npz")
    for T in (12, 40):
        taps = {}
        wav = ohifi.generator_forward(vsd, vhp, torch.from_numpy(g[f"mel_{T}"]), None, taps)
        np.testing.assert_allclose(wav[0, 0].numpy(), g[f"wav_{T}"], atol=TOL, rtol=0)
        if T == 12:
            for i in range(4):
                np.testing.assert_allclose(taps[f"ups.{i}"][0].numpy(), g[f"ups.{i}_12"], atol=2e-5, rtol=1e-5)
            np.testing.assert_allclose(taps["pre_tanh"][0].numpy(), g["pre_tanh_12"], atol=2e-5, rtol=1e-5)
    mel = torch.from_numpy(g["mel_40"])
    pre = ohifi.generator_forward(vsd, vhp, mel[:, :, :20])[0, 0].numpy()
    full = g["wav_40"][:20 * 320]
    # total look-ahead: 2 frames at every stage's input rate = 2*320 + 2*40 + 2*8 + 2*2 = 740 samples
    np.testing.assert_allclose(pre[:20 * 320 - 740], full[:20 * 320 - 740], atol=TOL, rtol=0)
    assert np.abs(pre - full).max() > 1e-3
    with pytest.raises(AssertionError):
        ohifi.generator_forward(vsd, vhp, mel[:, :, :4], {})      # no stateful streaming for this block
