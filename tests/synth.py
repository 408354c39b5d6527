"""Deterministic synthetic audio shared by tools/make_golden.py, the tests and bench.py."""
import numpy as np


def synth_audio(seed, n):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = 0.3 * np.sin(2 * np.pi * 440.0 * t) + 0.2 * np.sin(2 * np.pi * 1234.5 * t + 0.3)
    x = x * (0.5 + 0.5 * np.cos(2 * np.pi * 4.0 * t)) + rng.normal(0, 0.01, n)
    return (x / np.abs(x).max() * 0.8).astype(np.float32)
