"""Deterministic synthetic 16 kHz audio: the workload of bench.py, shared with tools/make_golden.py and the tests."""
import numpy as np


def synth_audio(seed, n):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = 0.3 * np.sin(2 * np.pi * 440.0 * t) + 0.2 * np.sin(2 * np.pi * 1234.5 * t + 0.3)
    x = x * (0.5 + 0.5 * np.cos(2 * np.pi * 4.0 * t)) + rng.normal(0, 0.01, n)
    return (x / np.abs(x).max() * 0.8).astype(np.float32)


def speechlike_audio(seconds, seed=1234, sr=16000):
    """BASELINE.md / SURVEY 8d synthetic workload: sum of 5 sinusoids with f ~ U(80, 3000) Hz
    re-drawn every 250 ms, 4 Hz raised-cosine envelope, + N(0, 0.01^2), peak 0.8."""
    rng = np.random.default_rng(seed)
    n = int(seconds * sr)
    seg = sr // 4
    nseg = (n + seg - 1) // seg
    t = np.arange(seg) / sr
    out = np.empty(nseg * seg, dtype=np.float32)
    for i in range(nseg):
        f = rng.uniform(80.0, 3000.0, 5)
        ph = rng.uniform(0, 2 * np.pi, 5)
        out[i * seg:(i + 1) * seg] = np.sin(2 * np.pi * f[:, None] * t[None, :] + ph[:, None]).sum(0)
    tt = np.arange(nseg * seg) / sr
    out *= (0.5 - 0.5 * np.cos(2 * np.pi * 4.0 * tt)).astype(np.float32)
    out += rng.normal(0, 0.01, out.shape).astype(np.float32)
    out = out[:n]
    return (out / np.abs(out).max() * 0.8).astype(np.float32)
