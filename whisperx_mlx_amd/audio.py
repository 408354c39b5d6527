"""Host-side audio constants and helpers mirroring /root/reference/whisperx/audio.py
(constants :13-22, pad_or_trim :68-91, mel_filters :94-109).  The log-mel itself runs
in HIP (csrc/logmel.hip) through WhisperHipEngine.logmel()."""
import numpy as np

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE      # 480000
N_FRAMES = N_SAMPLES // HOP_LENGTH          # 3000
N_SAMPLES_PER_TOKEN = HOP_LENGTH * 2
FRAMES_PER_SECOND = SAMPLE_RATE // HOP_LENGTH
TOKENS_PER_SECOND = SAMPLE_RATE // N_SAMPLES_PER_TOKEN   # 50


def pad_or_trim(array, length=N_SAMPLES, *, axis=-1):
    """audio.py:68-91 (numpy branch)."""
    array = np.asarray(array)
    if array.shape[axis] > length:
        array = array.take(indices=range(length), axis=axis)
    if array.shape[axis] < length:
        pad = [(0, 0)] * array.ndim
        pad[axis] = (0, length - array.shape[axis])
        array = np.pad(array, pad)
    return array


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filters(n_mels):
    """The (n_mels, 201) Slaney-normalised mel filterbank the reference ships as
    assets/mel_filters.npz (audio.py:94-109: librosa.filters.mel(sr=16000, n_fft=400,
    n_mels=n)), generated instead of copied; checked against the asset in tests."""
    assert n_mels in (80, 128), f"Unsupported n_mels: {n_mels}"
    fftfreqs = np.linspace(0.0, SAMPLE_RATE / 2, 1 + N_FFT // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(SAMPLE_RATE / 2), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels])
    return (weights * enorm[:, None]).astype(np.float32)


def split_chunks(audio, chunk_samples=N_SAMPLES):
    """Fixed 30 s windows, last one short (mlx_whisper_optimized_final.py:398-408)."""
    audio = np.asarray(audio, dtype=np.float32)
    return [audio[s: s + chunk_samples] for s in range(0, max(len(audio), 1), chunk_samples)]
