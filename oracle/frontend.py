"""Oracle of the mel front-end -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates `librosa_wav2spec` (utils/audio/__init__.py:37-84) as called by `StreamingVoiceConversion._wav_to_mel`
(inference/Conan.py:57-70) for an in-memory waveform with loud_norm=False (egs_bases/tts/dataset_params.yaml:15):

    x_stft = librosa.stft(wav, n_fft, hop_length, win_length, window='hann', pad_mode='constant')   # center=True
    mel    = librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) @ |x_stft|
    mel    = clip(log10(max(eps, mel)).T, mel_vmin, mel_vmax)

PIN: the arithmetic lives in librosa (requirements: librosa==0.9.2), which is neither installed in this image nor
vendored in the reference, so the reference's own output cannot be generated here.  This file restates librosa's
published algorithm -- periodic Hann window (scipy.signal.get_window(..., fftbins=True)), centred frames with zero
padding of n_fft//2, rfft, Slaney mel scale with 'slaney' area normalisation, float32 filterbank -- and
tests/test_oracle_frontend.py pins it against two independent public implementations that ARE in the image:
`transformers.audio_utils.{mel_filter_bank, window_function, spectrogram}` (documented as librosa-equivalent; filterbank
equal to 2e-9, log-mel to 1e-6) and `torch.stft` (magnitudes to 2e-7 of the peak), besides the explicit DFT sums /
triangle construction and analytic properties.  Against librosa itself it stays unpinned.
"""
import numpy as np
import scipy.signal

DEFAULTS = dict(fft_size=1024, hop_size=320, win_length=1024, num_mels=80, fmin=80, fmax=7600, sample_rate=16000,
                eps=1e-6, mel_vmin=-6.0, mel_vmax=1.5)          # egs/conan_emformer.yaml:31-38 + mel_vmin/vmax


def hz_to_mel(f):
    """librosa.core.convert.hz_to_mel(htk=False): Slaney's Auditory Toolbox scale."""
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm='slaney', dtype=float32) -> [n_mels, 1+n_fft//2]."""
    fftfreqs = np.linspace(0, float(sr) / 2, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis].astype(np.float32)
    return weights


def stft_mag(wav, n_fft, hop, win_length):
    """|librosa.stft(wav, n_fft, hop, win_length, window='hann', center=True, pad_mode='constant')| -> [1+n_fft//2, frames]."""
    wav = np.asarray(wav, dtype=np.float32)
    win = scipy.signal.get_window("hann", win_length, fftbins=True)
    if win_length < n_fft:                                   # librosa.util.pad_center
        lp = (n_fft - win_length) // 2
        win = np.pad(win, (lp, n_fft - win_length - lp))
    y = np.pad(wav, (n_fft // 2, n_fft // 2), mode="constant")
    n_frames = 1 + (len(y) - n_fft) // hop
    idx = np.arange(n_fft)[:, None] + hop * np.arange(n_frames)[None, :]
    frames = y[idx] * win[:, None].astype(np.float32)        # librosa multiplies in the input dtype (float32)
    spec = np.fft.rfft(frames, axis=0).astype(np.complex64)  # stft_matrix is complex64 for float32 input
    return np.abs(spec)


def wav2mel(wav, **kw):
    """-> mel [frames, n_mels] float32 = clip(librosa_wav2spec(wav)['mel'], mel_vmin, mel_vmax) (inference/Conan.py:57-70)."""
    p = dict(DEFAULTS); p.update(kw)
    mag = stft_mag(wav, p["fft_size"], p["hop_size"], p["win_length"])
    fmax = p["sample_rate"] / 2 if p["fmax"] == -1 else p["fmax"]
    fmin = 0 if p["fmin"] == -1 else p["fmin"]
    basis = mel_filterbank(p["sample_rate"], p["fft_size"], p["num_mels"], fmin, fmax)
    mel = basis @ mag
    mel = np.log10(np.maximum(p["eps"], mel))
    return np.clip(mel.T, p["mel_vmin"], p["mel_vmax"]).astype(np.float32)


def torch_mel_spectrogram(y, n_fft=1024, num_mels=80, sampling_rate=16000, hop_size=320, win_size=1024, fmin=80, fmax=None):
    """`mel_spectrogram` of the earlier loop (inference/Conan_previous.py:100-121), center=False, restated in numpy:
    reflect-pad (n_fft - hop) / 2 per side, frames from the padded signal's first sample, periodic Hann (torch.hann_window),
    sqrt(re^2 + im^2 + 1e-9), librosa mel basis (fmax None -> sr / 2), ln(clamp(., min=1e-5)).  y [samples] -> [num_mels, frames].
    PIN: the reference's function is torch.stft + F.pad, both in this image; tests/test_oracle_frontend.py compares."""
    y = np.asarray(y, dtype=np.float32)
    pad = int((n_fft - hop_size) / 2)
    yp = np.pad(y, (pad, pad), mode="reflect")
    win = scipy.signal.get_window("hann", win_size, fftbins=True)
    if win_size < n_fft:
        lp = (n_fft - win_size) // 2
        win = np.pad(win, (lp, n_fft - win_size - lp))
    n_frames = 1 + (len(yp) - n_fft) // hop_size
    idx = np.arange(n_fft)[:, None] + hop_size * np.arange(n_frames)[None, :]
    frames = yp[idx] * win[:, None].astype(np.float32)       # torch.stft multiplies input and window in float32
    spec = np.fft.rfft(frames.astype(np.float64), axis=0)
    mag = np.sqrt(spec.real ** 2 + spec.imag ** 2 + 1e-9).astype(np.float32)
    basis = mel_filterbank(sampling_rate, n_fft, num_mels, fmin, sampling_rate / 2 if fmax is None else fmax)
    return np.log(np.maximum(basis @ mag, 1e-5)).astype(np.float32)
