"""VAD chunk producer in front of the hot path: the reference's Silero front-end
(/root/reference/whisperx/vads/silero.py:15-66) and chunk merge (/root/reference/whisperx/vads/vad.py:20-53) as the
callable `HipWhisperPipeline` expects -- `vad(audio, chunk_size) -> [{"start", "end", "segments"}]`, every chunk at
most `chunk_size` seconds (whisperx/asr.py:122-143 `_segment_audio_with_vad`).

The Silero network itself stays a CPU pre-step outside the hot path, exactly as in the reference (a 4 MB JIT model at
318x real time, VAD_OPTIMIZATION_PLAN.md:31): this module only calls it.  It comes from `torch.hub`
(`snakers4/silero-vad`) -- from the hub cache or from a local clone (`repo_dir` / $WX_SILERO_DIR), since the GPU box has
no network.  Any other detector with Silero's `get_speech_timestamps` signature can be passed instead."""
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .audio import SAMPLE_RATE


def merge_chunks(segments: Sequence[Tuple[float, float]], chunk_size: float, onset: float = 0.5,
                 offset: Optional[float] = 0.363) -> List[Dict]:
    """vads/vad.py:20-53: merge speech turns [(start, end), ...] (seconds, in order) into chunks of at most
    `chunk_size` seconds; a turn that would push the running chunk past `chunk_size` opens a new one.  `onset` /
    `offset` are accepted for signature compatibility (the reference does not use them in the merge either)."""
    if len(segments) == 0:              # vads/silero.py:60-63
        print("No active speech found in audio")
        return []
    curr_end = 0
    merged, seg_idxs = [], []
    curr_start = segments[0][0]
    for (s, e) in segments:
        if e - curr_start > chunk_size and curr_end - curr_start > 0:
            merged.append({"start": curr_start, "end": curr_end, "segments": seg_idxs})
            curr_start = s
            seg_idxs = []
        curr_end = e
        seg_idxs.append((s, e))
    merged.append({"start": curr_start, "end": curr_end, "segments": seg_idxs})
    return merged


class SileroVad:
    """vads/silero.py: `get_speech_timestamps(waveform, model=..., sampling_rate=16000, max_speech_duration_s=chunk_size,
    threshold=vad_onset)` -> turns in seconds -> merge_chunks.  Call: vad(audio float32 mono 16 kHz, chunk_size)."""

    def __init__(self, get_speech_timestamps: Callable, model=None, vad_onset: float = 0.5, vad_offset: Optional[float] = 0.363):
        if not (0 < vad_onset < 1):                      # vads/vad.py:9-13
            raise ValueError("vad_onset is a decimal value between 0 and 1.")
        self.get_speech_timestamps = get_speech_timestamps
        self.model = model
        self.vad_onset, self.vad_offset = vad_onset, vad_offset

    @classmethod
    def from_hub(cls, repo_dir: Optional[str] = None, vad_onset: float = 0.5, vad_offset: Optional[float] = 0.363):
        """vads/silero.py:23-28.  repo_dir (or $WX_SILERO_DIR): a local clone of snakers4/silero-vad; otherwise the
        torch.hub cache must already hold it (no network on the GPU box)."""
        import torch
        print(">>Performing voice activity detection using Silero...")
        repo_dir = repo_dir or os.environ.get("WX_SILERO_DIR")
        try:
            if repo_dir:
                model, utils = torch.hub.load(repo_or_dir=repo_dir, model="silero_vad", source="local", onnx=False, trust_repo=True)
            else:
                model, utils = torch.hub.load(repo_or_dir="snakers4/silero-vad", model="silero_vad", force_reload=False,
                                              onnx=False, trust_repo=True)
        except Exception as e:           # noqa: BLE001
            raise RuntimeError("Silero VAD is not available offline: pass repo_dir= (or set WX_SILERO_DIR) to a local clone of "
                               f"snakers4/silero-vad, or use vad_method='none' for fixed windows ({e})") from e
        return cls(utils[0], model, vad_onset, vad_offset)

    def speech_turns(self, audio: np.ndarray, chunk_size: float) -> List[Tuple[float, float]]:
        import torch
        wav = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32))
        ts = self.get_speech_timestamps(wav, model=self.model, sampling_rate=SAMPLE_RATE, max_speech_duration_s=chunk_size,
                                        threshold=self.vad_onset)
        return [(t["start"] / SAMPLE_RATE, t["end"] / SAMPLE_RATE) for t in ts]

    def __call__(self, audio: np.ndarray, chunk_size: float = 30) -> List[Dict]:
        assert chunk_size > 0
        return merge_chunks(self.speech_turns(audio, chunk_size), chunk_size, onset=self.vad_onset, offset=self.vad_offset)
