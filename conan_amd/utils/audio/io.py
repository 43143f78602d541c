"""utils/audio/io.py:7-13 of the reference: 16-bit PCM wav writer (mp3 transcoding through ffmpeg is not provided)."""
import numpy as np
from scipy.io import wavfile


def save_wav(wav, path, sr, norm=False):
    if path[-4:] == ".mp3":
        raise NotImplementedError("mp3 output needs ffmpeg (utils/audio/io.py:16-22); write .wav")
    wav = np.asarray(wav, dtype=np.float32)
    if norm:
        wav = wav / np.abs(wav).max()
    wav = wav * 32767
    wavfile.write(path[:-4] + ".wav", sr, wav.astype(np.int16))
