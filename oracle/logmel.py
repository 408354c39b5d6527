"""Oracle: log-mel spectrogram.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement (numpy) of the reference's ``log_mel_spectrogram``
(/root/reference/whisperx/audio.py:112-159) and ``pad_or_trim`` (:68-91).
Pinned by tests/golden/logmel_*.npz, which were produced by importing the
reference's own audio.py (tools/make_golden.py).
"""
import numpy as np

SAMPLE_RATE = 16000      # audio.py:13
N_FFT = 400              # audio.py:14
HOP_LENGTH = 160         # audio.py:15
N_SAMPLES = 480000       # audio.py:17
N_FRAMES = 3000          # audio.py:18


def pad_or_trim(array, length=N_SAMPLES, axis=-1):
    """audio.py:68-91 (numpy branch): zero-pad on the right or trim."""
    array = np.asarray(array)
    if array.shape[axis] > length:
        array = array.take(indices=range(length), axis=axis)
    if array.shape[axis] < length:
        pad = [(0, 0)] * array.ndim
        pad[axis] = (0, length - array.shape[axis])
        array = np.pad(array, pad)
    return array


def hann_window_periodic(n=N_FFT):
    """torch.hann_window(N_FFT) (periodic=True default), audio.py:149."""
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)).astype(np.float32)


def stft_power(audio):
    """|STFT|^2 with torch.stft defaults (center=True, reflect pad n_fft//2),
    last frame dropped -- audio.py:150-151.  Returns (201, n_frames) float32."""
    x = np.asarray(audio, dtype=np.float32)
    pad = N_FFT // 2
    xp = np.pad(x, (pad, pad), mode="reflect")
    n_frames = 1 + (len(xp) - N_FFT) // HOP_LENGTH
    idx = np.arange(N_FFT)[None, :] + HOP_LENGTH * np.arange(n_frames)[:, None]
    frames = xp[idx] * hann_window_periodic()[None, :]
    spec = np.fft.rfft(frames.astype(np.float64), axis=1)        # (n_frames, 201)
    power = (spec.real ** 2 + spec.imag ** 2).astype(np.float32)
    return power[:-1].T                                           # drop last frame (:151)


def log_mel_spectrogram(audio, filters, padding=0):
    """audio.py:112-159.  ``filters`` is the (n_mels, 201) float32 filterbank
    (audio.py:94-109).  Returns (n_mels, n_frames) float32; the dynamic-range
    clamp uses the max over the WHOLE array passed in (audio.py:157)."""
    x = np.asarray(audio, dtype=np.float32)
    if padding > 0:
        x = np.pad(x, (0, padding))
    mag = stft_power(x)
    mel = filters.astype(np.float32) @ mag
    log_spec = np.log10(np.maximum(mel, 1e-10))
    log_spec = np.maximum(log_spec, log_spec.max() - 8.0)
    return ((log_spec + 4.0) / 4.0).astype(np.float32)


def log_mel_chunks(pcm, n_valid, filters):
    """Batched form used by path C (mlx_whisper_optimized_final.py:428-434):
    each 30 s chunk is zero-padded to 480000 samples, log-mel'd on its own
    (per-chunk max) and laid out channels-last (B, 3000, n_mels)."""
    out = []
    for row, n in zip(pcm, n_valid):
        x = pad_or_trim(np.asarray(row[:n], dtype=np.float32), N_SAMPLES)
        out.append(log_mel_spectrogram(x, filters).T)
    return np.stack(out)
