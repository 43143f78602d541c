"""The reference's wav -> mel front-end interface (utils/audio/__init__.py:37-84) on the HIP path.

`librosa_wav2spec` keeps the reference's keyword arguments; the STFT, mel projection and log run on the GPU
(conan_wav2mel).  A path argument is read with the standard-library `wave` module (PCM wav at the configured sample
rate: the reference resamples through librosa.core.load, which is not re-implemented); loud_norm / trim_long_sil (pyloudnorm /
webrtcvad, both off in egs/conan_emformer.yaml) raise NotImplementedError."""
import wave

import numpy as np
import torch


def load_wav(path, sample_rate):
    """PCM wav file -> float32 mono in [-1, 1]; the file's rate must equal `sample_rate`."""
    with wave.open(path, "rb") as f:
        sr, ch, width, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
        raw = f.readframes(n)
    if sr != sample_rate:
        raise ValueError(f"{path}: sample rate {sr} != {sample_rate} (resampling is outside this front-end)")
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"{path}: unsupported sample width {width}")
    return x.reshape(-1, ch).mean(1) if ch > 1 else x


def librosa_pad_lr(x, fsize, fshift, pad_sides=1):
    """utils/audio/__init__.py:13-22."""
    assert pad_sides in (1, 2)
    pad = (x.shape[0] // fshift + 1) * fshift - x.shape[0]
    if pad_sides == 1:
        return 0, pad
    return pad // 2, pad // 2 + pad % 2


def librosa_wav2spec(wav_path, fft_size=1024, hop_size=256, win_length=1024, window="hann", num_mels=80, fmin=80, fmax=-1,
                     eps=1e-6, sample_rate=22050, loud_norm=False, trim_long_sil=False, ctx=None):
    """Same contract as the reference for the keys the inference path reads: {'wav', 'mel' [T, num_mels], 'wav_orig'}
    ('linear' / 'mel_basis' are not produced).  `ctx`: a finalized conan_amd.runtime.Context (device + library)."""
    if loud_norm or trim_long_sil:
        raise NotImplementedError("loud_norm / trim_long_sil are off on the inference path (egs_bases/tts/dataset_params.yaml:15)")
    if window != "hann":
        raise NotImplementedError("only the Hann window of the reference configuration")
    if ctx is None:
        raise ValueError("librosa_wav2spec needs ctx= (a finalized conan_amd.runtime.Context): the transform runs on the GPU")
    wav = load_wav(wav_path, sample_rate) if isinstance(wav_path, str) else np.asarray(wav_path, dtype=np.float32)
    wav_orig = np.copy(wav)
    mel = ctx.wav2mel(torch.from_numpy(wav), fft_size=fft_size, hop_size=hop_size, win_length=win_length, num_mels=num_mels,
                      fmin=fmin, fmax=fmax, sample_rate=sample_rate, eps=eps, mel_vmin=-1e30, mel_vmax=1e30)[0].cpu().numpy()
    l_pad, r_pad = librosa_pad_lr(wav, fft_size, hop_size, 1)
    wav = np.pad(wav, (l_pad, r_pad), mode="constant", constant_values=0.0)[:mel.shape[0] * hop_size]
    return {"wav": wav, "mel": mel, "wav_orig": wav_orig}


def mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False, ctx=None):
    """The earlier loop's front-end (`mel_spectrogram` nested in inference/Conan_previous.py:100-121) on the HIP path:
    reflect padding of (n_fft - hop_size) / 2 per side, torch.stft(center=False) with a periodic Hann window,
    sqrt(re^2 + im^2 + 1e-9), librosa (Slaney) mel basis with fmax=None -> Nyquist, ln(clamp(., min=1e-5)).
    y: [B, samples] tensor in [-1, 1] -> [B, num_mels, samples // hop_size] (the reference's layout)."""
    if center:
        raise NotImplementedError("the reference calls it with center=False only (inference/Conan_previous.py:133)")
    if ctx is None:
        raise ValueError("mel_spectrogram needs ctx= (a finalized conan_amd.runtime.Context): the transform runs on the GPU")
    y = torch.as_tensor(y, dtype=torch.float32)
    mel = ctx.wav2mel(y, fft_size=n_fft, hop_size=hop_size, win_length=win_size, num_mels=num_mels, fmin=fmin,
                      fmax=-1 if fmax is None else fmax, sample_rate=sampling_rate, eps=1e-5, mel_vmin=-1e30, mel_vmax=1e30,
                      framing=1, natural_log=True, mag_eps=1e-9)
    return mel.transpose(1, 2)
