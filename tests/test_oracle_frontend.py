"""The mel front-end oracle (oracle/frontend.py) restates librosa's published algorithm; librosa is not installed here
and the reference holds no fixture for it, so it is pinned against two independent public implementations in the image
(transformers.audio_utils, torch.stft), an independently written formulation and analytic properties."""
import numpy as np
import pytest

from oracle import frontend as fe


def _wave(seconds=0.4, sr=16000, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(int(seconds * sr)) / sr
    x = 0.4 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 2300 * t + 1.0) + 0.05 * rng.standard_normal(t.shape)
    return x.astype(np.float32)


def test_stft_matches_explicit_dft_sums():
    x = _wave(0.1)
    n_fft, hop = 256, 80
    mag = fe.stft_mag(x, n_fft, hop, n_fft)
    assert mag.shape == (n_fft // 2 + 1, 1 + len(x) // hop)
    # independent formulation: explicit zero-padded frames and a dense DFT matrix in float64
    k = np.arange(n_fft)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * k / n_fft)                      # periodic Hann
    y = np.concatenate([np.zeros(n_fft // 2), x.astype(np.float64), np.zeros(n_fft // 2)])
    dft = np.exp(-2j * np.pi * np.outer(np.arange(n_fft // 2 + 1), k) / n_fft)
    for f in (0, 1, 7, mag.shape[1] - 1):
        ref = np.abs(dft @ (y[f * hop:f * hop + n_fft] * win))
        np.testing.assert_allclose(mag[:, f], ref, rtol=1e-4, atol=1e-4)


def test_mel_filterbank_properties():
    fb = fe.mel_filterbank(16000, 1024, 80, 80, 7600)
    assert fb.shape == (80, 513) and fb.dtype == np.float32 and (fb >= 0).all()
    freqs = np.linspace(0, 8000, 513)
    mel_f = fe.mel_to_hz(np.linspace(fe.hz_to_mel(80), fe.hz_to_mel(7600), 82))
    for i in (0, 10, 40, 79):
        nz = np.nonzero(fb[i])[0]
        assert freqs[nz[0]] > mel_f[i] and freqs[nz[-1]] < mel_f[i + 2]           # support = (left edge, right edge)
        assert abs(freqs[np.argmax(fb[i])] - mel_f[i + 1]) <= 8000 / 512          # peak at the centre frequency
        # slaney normalisation: triangle of height 2/(right-left) -> unit area (bin width 15.625 Hz)
        assert abs(fb[i].sum() * (8000 / 512) - 1.0) < 0.08
    # Slaney scale: linear below 1 kHz, logarithmic above, continuous at 1 kHz
    assert abs(fe.hz_to_mel(1000.0) - 15.0) < 1e-12 and abs(fe.mel_to_hz(fe.hz_to_mel(3210.0)) - 3210.0) < 1e-9


def test_wav2mel_shape_clip_and_tone_location():
    x = _wave(0.5)
    mel = fe.wav2mel(x)
    assert mel.shape == (1 + len(x) // 320, 80) and mel.dtype == np.float32
    assert mel.min() >= -6.0 and mel.max() <= 1.5
    fb_centres = fe.mel_to_hz(np.linspace(fe.hz_to_mel(80), fe.hz_to_mel(7600), 82))[1:-1]
    loud = int(np.argmax(mel[10]))
    assert abs(fb_centres[loud] - 440) < 60                                       # the 440 Hz partial dominates
    assert np.all(fe.wav2mel(np.zeros(3200, np.float32)) == -6.0)                 # silence sits on the floor


def _signal(seed=3, dur=1.3, extra=57):
    rng = np.random.default_rng(seed)
    t = np.arange(int(dur * 16000) + extra) / 16000
    return (0.5 * np.sin(2 * np.pi * 220 * t) * np.exp(-t) + 0.1 * np.sin(2 * np.pi * 3100 * t) + 0.02 * rng.standard_normal(t.shape)).astype(np.float32)


def test_pinned_against_transformers_audio_utils():
    """Independent public implementation #1: transformers.audio_utils (documented as librosa-compatible) - Slaney
    filterbank, periodic Hann, centred zero-padded amplitude spectrogram, log10 with floor."""
    au = pytest.importorskip("transformers.audio_utils")
    p = fe.DEFAULTS
    fb = au.mel_filter_bank(num_frequency_bins=1 + p["fft_size"] // 2, num_mel_filters=p["num_mels"], min_frequency=p["fmin"],
                            max_frequency=p["fmax"], sampling_rate=p["sample_rate"], norm="slaney", mel_scale="slaney")
    mine = fe.mel_filterbank(p["sample_rate"], p["fft_size"], p["num_mels"], p["fmin"], p["fmax"])
    assert fb.shape == mine.T.shape and np.abs(fb.T - mine).max() < 1e-8
    # (the pure tone has mel bins 6 decades below its peak: there the oracle carries librosa's complex64 rounding, the
    # transformers routine computes in float64 - 1e-3 in log10 units covers that noise floor)
    for wav, tol in ((_signal(), 1e-5), (_signal(5, 0.4, 0), 1e-5), ((0.9 * np.sin(2 * np.pi * 440 * np.arange(9000) / 16000)).astype(np.float32), 1e-3)):
        win = au.window_function(p["win_length"], "hann", periodic=True)
        spec = au.spectrogram(wav, win, frame_length=p["win_length"], hop_length=p["hop_size"], fft_length=p["fft_size"], power=1.0, center=True,
                              pad_mode="constant", onesided=True, mel_filters=fb, mel_floor=p["eps"], log_mel="log10")
        ref = np.clip(spec.T, p["mel_vmin"], p["mel_vmax"])
        got = fe.wav2mel(wav)
        assert got.shape == ref.shape and np.abs(got - ref).max() < tol, np.abs(got - ref).max()


def test_stft_pinned_against_torch_stft():
    """Independent public implementation #2: torch.stft(center=True, pad_mode='constant', periodic Hann)."""
    import torch
    wav = _signal()
    S = torch.stft(torch.from_numpy(wav), 1024, hop_length=320, win_length=1024, window=torch.hann_window(1024, periodic=True), center=True,
                   pad_mode="constant", return_complex=True).abs().numpy()
    mag = fe.stft_mag(wav, 1024, 320, 1024)
    assert S.shape == mag.shape and np.abs(S - mag).max() < 1e-6 * mag.max()


def test_torch_stft_frontend_of_the_earlier_loop_pinned_against_torch():
    """inference/Conan_previous.py:100-121 is torch.nn.functional.pad(reflect) + torch.stft(center=False) + a librosa mel
    basis + log(clamp): the same torch calls, on the mel basis pinned above, are the independent implementation here."""
    import torch
    for wav in (_signal(), _signal(5, 0.4, 0)):
        y = torch.from_numpy(wav)[None]
        yp = torch.nn.functional.pad(y.unsqueeze(1), (int((1024 - 320) / 2), int((1024 - 320) / 2)), mode="reflect").squeeze(1)
        spec = torch.view_as_real(torch.stft(yp, 1024, hop_length=320, win_length=1024, window=torch.hann_window(1024), center=False,
                                             pad_mode="reflect", normalized=False, onesided=True, return_complex=True))
        spec = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)
        basis = torch.from_numpy(fe.mel_filterbank(16000, 1024, 80, 80, 8000.0))
        ref = torch.log(torch.clamp(torch.matmul(basis, spec), min=1e-5))[0].numpy()
        got = fe.torch_mel_spectrogram(wav)
        assert got.shape == ref.shape == (80, len(wav) // 320)
        assert np.abs(got - ref).max() < 2e-4, np.abs(got - ref).max()      # float32 FFT noise under a natural log
