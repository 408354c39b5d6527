"""Long-form host chunker: the reference's `BatchProcessor`
(/root/reference/whisperx/batch_processor.py:25-338) behind the same class/method names, with
the per-batch step actually batched on the GPU (the reference pads a batch and then loops
`mlx_whisper.transcribe` per chunk, :150-184)."""
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

from .audio import SAMPLE_RATE


@dataclass
class AudioChunk:
    """batch_processor.py:16-22"""
    audio: np.ndarray
    start_time: float
    end_time: float
    segment_idx: int


class BatchProcessor:
    def __init__(self, batch_size: int = 8, chunk_duration: float = 30.0, overlap: float = 0.5):
        self.batch_size = batch_size
        self.chunk_duration = chunk_duration
        self.overlap = overlap
        self.chunk_samples = int(chunk_duration * SAMPLE_RATE)
        self.overlap_samples = int(overlap * SAMPLE_RATE)

    def create_chunks(self, audio: np.ndarray, segments: List[Dict[str, float]]) -> List[AudioChunk]:
        """:47-99 -- a segment no longer than chunk_duration is one chunk; a longer one becomes
        ceil(duration / (chunk_duration - overlap)) windows that start (chunk_duration - overlap) apart."""
        out: List[AudioChunk] = []
        hop = self.chunk_duration - self.overlap
        for idx, seg in enumerate(segments):
            t0, t1 = seg["start"], seg["end"]
            if t1 - t0 <= self.chunk_duration:
                spans = [(t0, t1)]
            else:
                n = math.ceil((t1 - t0) / hop)
                spans = [(t0 + i * hop, min(t0 + i * hop + self.chunk_duration, t1)) for i in range(n)]
            for a, b in spans:
                out.append(AudioChunk(audio[int(a * SAMPLE_RATE): int(b * SAMPLE_RATE)], a, b, idx))
        return out

    def create_batches(self, chunks: List[AudioChunk]) -> List[List[AudioChunk]]:
        """:101-116"""
        return [chunks[i: i + self.batch_size] for i in range(0, len(chunks), self.batch_size)]

    def pad_batch(self, batch: List[AudioChunk]) -> Tuple[np.ndarray, List[int]]:
        """:118-148 -- zero-pad on the right to the longest chunk of the batch."""
        lengths = [len(c.audio) for c in batch]
        width = max(lengths)
        padded = np.zeros((len(batch), width), dtype=batch[0].audio.dtype)
        for row, c in zip(padded, batch):
            row[: len(c.audio)] = c.audio
        return padded, lengths

    def process_batch(self, batch: List[AudioChunk], backend, language: Optional[str] = "en",
                      task: str = "transcribe") -> List[Dict]:
        """replaces process_batch_with_mlx (:150-184): one batched GPU pass over the chunks."""
        res = backend._decode_chunks([np.asarray(c.audio, dtype=np.float32) for c in batch], language, task, False)
        return [{"text": r["text"], "tokens": r["tokens"], "language": r["language"]} for r in res]

    def merge_results(self, chunks: List[AudioChunk], results: List[Dict], segments: List[Dict]) -> List[Dict]:
        """:186-241 -- one output segment per input segment; multi-chunk segments are stitched."""
        by_seg: Dict[int, List[Tuple[AudioChunk, Dict]]] = {}
        for c, r in zip(chunks, results):
            by_seg.setdefault(c.segment_idx, []).append((c, r))
        final = []
        for idx, seg in enumerate(segments):
            parts = by_seg.get(idx)
            if not parts:
                text = ""
            elif len(parts) == 1:
                text = parts[0][1].get("text", "").strip()
            else:
                text = self._merge_overlapping_text(parts)
            final.append({"start": seg["start"], "end": seg["end"], "text": text})
        return final

    def _merge_overlapping_text(self, chunk_results: List[Tuple[AudioChunk, Dict]]) -> str:
        """:243-276 -- first chunk verbatim; every later chunk drops its first fifth of words
        (the presumed overlap) when it has more than five."""
        if not chunk_results:
            return ""
        ordered = sorted(chunk_results, key=lambda cr: cr[0].start_time)
        merged = ordered[0][1].get("text", "").strip()
        for _c, r in ordered[1:]:
            words = r.get("text", "").strip().split()
            if len(words) > 5:
                words = words[len(words) // 5:]
            tail = " ".join(words)
            if tail:
                merged += " " + tail
        return merged


def batch_transcribe(audio: np.ndarray, segments: List[Dict], backend, batch_size: int = 8, chunk_duration: float = 30.0,
                     decode_options: Optional[Dict] = None, print_progress: bool = False) -> List[Dict]:
    """:279-338 with the model path replaced by a WhisperHipBackend."""
    proc = BatchProcessor(batch_size=batch_size, chunk_duration=chunk_duration, overlap=0.5)
    chunks = proc.create_chunks(audio, segments)
    if print_progress:
        print(f"Created {len(chunks)} chunks from {len(segments)} segments")
    batches = proc.create_batches(chunks)
    if print_progress:
        print(f"Processing {len(batches)} batches of size {batch_size}")
    opts = decode_options or {}
    # all batches go to the backend in one call: its scheduler cuts the chunk list into passes of `batch_size` rows
    # (times the backend's `coalesce`) and keeps several passes in flight, instead of the reference's batch-after-batch
    # loop (:318-327); results come back in chunk order
    res = backend._decode_chunks([np.asarray(c.audio, dtype=np.float32) for c in chunks], opts.get("language", "en"),
                                 opts.get("task", "transcribe"), False, forced_len=int(opts.get("forced_len", 0)),
                                 rows_per_pass=None if getattr(backend, "auto_rows", False)
                                 else batch_size * getattr(backend, "coalesce", 1)) if chunks else []
    results = [{"text": r["text"], "tokens": r["tokens"], "language": r["language"]} for r in res]
    return proc.merge_results(chunks, results, segments)
