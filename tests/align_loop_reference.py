"""TEST INFRASTRUCTURE: the char -> word -> sentence assembly of whisperx_mlx_amd/alignment.py as it stood before it was made
linear in the number of characters (round 4) -- one dict per character, filtered per sentence and per word the way the
reference filters its pandas frame (/root/reference/whisperx/alignment.py:296-343).  tests/test_align_host.py holds the
shipped assembly against this one on random texts and paths; both are pinned to the reference's own output by
tests/golden/align.json."""
import math
from typing import List

import numpy as np

from whisperx_mlx_amd.alignment import (LANGUAGES_WITHOUT_SPACES, SAMPLE_RATE, _nanmax, _nanmean, _nanmin, interpolate_nans,
                                        sentence_spans)


def merge_repeats(path_tok, path_score, transcript):
    """alignment.py:597-613 on the device path arrays -> [(label, start, end, score)]."""
    segs, i1, n = [], 0, len(path_tok)
    while i1 < n:
        i2 = i1
        while i2 < n and path_tok[i1] == path_tok[i2]:
            i2 += 1
        score = sum(float(path_score[k]) for k in range(i1, i2)) / (i2 - i1)
        segs.append((transcript[path_tok[i1]], i1, i2, score))
        i1 = i2
    return segs


def align_batch(
    items,
    model,
    align_model_metadata: dict,
    device: str,
    interpolate_method: str = "nearest",
    return_char_alignments: bool = False,
    print_progress: bool = False,
    combined_progress: bool = False,
    _aligner=None,
    _sentence_spans=None,
    _trace=None,
) -> List[dict]:
    """align() for many (transcript, audio) pairs at once: one result dict per pair, each identical to align() of that pair.

    The reference aligns one VAD segment's transcript per call, one wav2vec2 forward per transcript segment
    (/root/reference/whisperx/backends/mlx_lightning.py:290-369 drives /root/reference/whisperx/alignment.py:206-258, with
    its "TODO batched inference").  Every segment is independent, so here the segments of ALL pairs go through the
    aligner together (sorted by length, 64 per forward) -- which is also what lets a rank align its whole shard of a
    job at once (parallel.transcribe_batch_sharded).

    `_trace`: a list that receives, per pair, the structure the multi-GPU record carries (parallel.pack_aligned)."""
    model_dictionary = align_model_metadata["dictionary"]
    model_lang = align_model_metadata["language"]
    model_type = align_model_metadata["type"]
    _aligner_takes_device = _aligner is None          # injected (CPU test) aligners get numpy, as the reference's model does
    if _aligner is None:
        if model_type != "hip":
            raise NotImplementedError(f"Align model of type {model_type} not supported.")
        raise ValueError("the loop statement needs an injected aligner")
    span_fn = _sentence_spans or (lambda sdx, text: sentence_spans(text))
    blank_id = 0
    for char, code in model_dictionary.items():
        if char == '[pad]' or char == '<pad>':
            blank_id = code

    prepared = []
    jobs = []                # over all pairs: (pair, sdx, waveform, tokens, text_clean)
    for pi, (transcript, audio) in enumerate(items):
        is_dev = False
        try:
            import torch
            if torch.is_tensor(audio):
                # audio resident in HBM stays there (the default aligner builds its batches on the device); CPU tensors
                # and everything else become numpy as in the reference
                is_dev = audio.is_cuda and _aligner_takes_device
                if not is_dev:
                    audio = audio.detach().cpu().numpy()
        except ImportError:       # pragma: no cover
            pass
        if isinstance(audio, str):
            from .backend import load_audio
            audio = load_audio(audio)
        if not is_dev:
            audio = np.asarray(audio, dtype=np.float32)
        if audio.ndim == 2:
            audio = audio[0]
        MAX_DURATION = audio.shape[0] / SAMPLE_RATE
        transcript = list(transcript)
        total_segments = len(transcript)
        segment_data = {}
        # 1. Preprocess to keep only characters in dictionary (alignment.py:140-201)
        for sdx, segment in enumerate(transcript):
            if print_progress:
                base_progress = ((sdx + 1) / total_segments) * 100
                percent_complete = (50 + base_progress / 2) if combined_progress else base_progress
                print(f"Progress: {percent_complete:.2f}%...")
            text = segment["text"]
            num_leading = len(text) - len(text.lstrip())
            num_trailing = len(text) - len(text.rstrip())
            clean_char, clean_cdx = [], []
            for cdx, char in enumerate(text):
                char_ = char.lower()
                if model_lang not in LANGUAGES_WITHOUT_SPACES:
                    char_ = char_.replace(" ", "|")
                if cdx < num_leading:
                    pass
                elif cdx > len(text) - num_trailing - 1:
                    pass
                elif char_ in model_dictionary.keys():
                    clean_char.append(char_)
                    clean_cdx.append(cdx)
                else:
                    clean_char.append('*')
                    clean_cdx.append(cdx)
            segment_data[sdx] = {"clean_char": clean_char, "clean_cdx": clean_cdx,
                                 "sentence_spans": list(span_fn(sdx, text))}
        # 2a. which segments can be aligned; their waveforms join the batch (alignment.py:206-249)
        for sdx, segment in enumerate(transcript):
            t1, t2 = segment["start"], segment["end"]
            if len(segment_data[sdx]["clean_char"]) == 0 or t1 >= MAX_DURATION:
                continue
            text_clean = "".join(segment_data[sdx]["clean_char"])
            tokens = [model_dictionary.get(c, -1) for c in text_clean]
            f1, f2 = int(t1 * SAMPLE_RATE), int(t2 * SAMPLE_RATE)
            jobs.append((pi, sdx, audio[f1:f2], tokens, text_clean))
        prepared.append((transcript, segment_data, MAX_DURATION))

    results = _aligner([j[2] for j in jobs], [j[3] for j in jobs], blank_id, 2) if jobs else []
    by_key = {(j[0], j[1]): (r, j[4]) for j, r in zip(jobs, results)}
    out = []
    for pi, (transcript, segment_data, MAX_DURATION) in enumerate(prepared):
        trace = [] if _trace is not None else None
        out.append(_assemble(pi, transcript, segment_data, MAX_DURATION, by_key, model_lang, interpolate_method,
                             return_char_alignments, trace))
        if _trace is not None:
            _trace.append(trace)
    return out


def _assemble(pi, transcript, segment_data, MAX_DURATION, by_key, model_lang, interpolate_method, return_char_alignments, trace):
    """char -> word -> sentence assembly of one (transcript, audio) pair (alignment.py:206-380 behind the emissions).
    `trace` (when not None) receives one entry per OUTPUT segment: ("fail", sdx) for a segment returned unaligned, or
    ("ok", sdx, [sentence spans (begin, end) of the sentences the output segment joins])."""
    aligned_segments: List[dict] = []
    for sdx, segment in enumerate(transcript):
        t1, t2, text = segment["start"], segment["end"], segment["text"]
        aligned_seg = {"start": t1, "end": t2, "text": text, "words": [], "chars": None}
        if return_char_alignments:
            aligned_seg["chars"] = []
        if len(segment_data[sdx]["clean_char"]) == 0:
            print(f'Failed to align segment ("{segment["text"]}"): no characters in this segment found in model dictionary, resorting to original...')
            aligned_segments.append(aligned_seg)
            if trace is not None:
                trace.append(("fail", sdx))
            continue
        if t1 >= MAX_DURATION:
            print(f'Failed to align segment ("{segment["text"]}"): original start time longer than audio duration, skipping...')
            aligned_segments.append(aligned_seg)
            if trace is not None:
                trace.append(("fail", sdx))
            continue
        (n_frames, path_tok, path_score), text_clean = by_key[(pi, sdx)]
        if path_tok is None or n_frames < 2:
            print(f'Failed to align segment ("{segment["text"]}"): backtrack failed, resorting to original...')
            aligned_segments.append(aligned_seg)
            if trace is not None:
                trace.append(("fail", sdx))
            continue
        char_segments = merge_repeats(path_tok, path_score, text_clean)
        duration = t2 - t1
        ratio = duration * 1 / (n_frames - 1)

        # assign timestamps to aligned characters (alignment.py:281-309)
        clean_cdx = segment_data[sdx]["clean_cdx"]
        cdx_pos = {c: i for i, c in enumerate(clean_cdx)}
        rows = []
        word_idx = 0
        for cdx, char in enumerate(text):
            start = end = score = None
            if cdx in cdx_pos:
                _lab, s0, e0, sc = char_segments[cdx_pos[cdx]]
                start = round(s0 * ratio + t1, 3)
                end = round(e0 * ratio + t1, 3)
                score = round(sc, 3)
            rows.append({"char": char, "start": start, "end": end, "score": score, "word-idx": word_idx})
            if model_lang in LANGUAGES_WITHOUT_SPACES:
                word_idx += 1
            elif cdx == len(text) - 1 or text[cdx + 1] == " ":
                word_idx += 1

        aligned_subsegments = []
        for sstart, send in segment_data[sdx]["sentence_spans"]:
            curr = rows[sstart: send + 1]          # pandas .loc is end-inclusive (alignment.py:317)
            sentence_text = text[sstart:send]
            sentence_start = _nanmin(r["start"] for r in curr)
            sentence_end = _nanmax(r["end"] for r in curr if r["char"] != ' ')
            sentence_words = []
            seen = []
            for r in curr:
                if r["word-idx"] not in seen:
                    seen.append(r["word-idx"])
            for widx in seen:
                wchars = [r for r in curr if r["word-idx"] == widx]
                word_text = "".join(r["char"] for r in wchars).strip()
                if len(word_text) == 0:
                    continue
                wchars = [r for r in wchars if r["char"] != " "]
                word_start = _nanmin(r["start"] for r in wchars)
                word_end = _nanmax(r["end"] for r in wchars)
                # pandas .mean() yields np.float64, whose round() is numpy's (scale, rint, unscale)
                word_score = float(round(np.float64(_nanmean(r["score"] for r in wchars)), 3))
                word_segment = {"word": word_text}
                if not math.isnan(word_start):
                    word_segment["start"] = word_start
                if not math.isnan(word_end):
                    word_segment["end"] = word_end
                if not math.isnan(word_score):
                    word_segment["score"] = word_score
                sentence_words.append(word_segment)
            sub = {"text": sentence_text, "start": sentence_start, "end": sentence_end, "words": sentence_words,
                   "_span": (sstart, send)}
            if return_char_alignments:
                chars = []
                for r in curr:
                    c = {"char": r["char"]}
                    for key in ("start", "end", "score"):
                        if r[key] is not None and r[key] != -1:
                            c[key] = r[key]
                    chars.append(c)
                sub["chars"] = chars
            aligned_subsegments.append(sub)

        if aligned_subsegments:
            starts = interpolate_nans([s["start"] for s in aligned_subsegments], method=interpolate_method)
            ends = interpolate_nans([s["end"] for s in aligned_subsegments], method=interpolate_method)
            for s, a, b in zip(aligned_subsegments, starts, ends):
                s["start"], s["end"] = a, b
            # concatenate sentences with same timestamps; groupby sorts by (start, end) and
            # drops NaN keys (alignment.py:364-372)
            groups = {}
            for s in aligned_subsegments:
                if math.isnan(s["start"]) or math.isnan(s["end"]):
                    continue
                groups.setdefault((s["start"], s["end"]), []).append(s)
            joiner = "".join if model_lang in LANGUAGES_WITHOUT_SPACES else " ".join
            for key in sorted(groups):
                grp = groups[key]
                rec = {"start": key[0], "end": key[1], "text": joiner(g["text"] for g in grp),
                       "words": [w for g in grp for w in g["words"]]}
                if return_char_alignments:
                    rec["chars"] = [c for g in grp for c in g["chars"]]
                aligned_segments.append(rec)
                if trace is not None:
                    trace.append(("ok", sdx, [g["_span"] for g in grp]))

    word_segments: List[dict] = []
    for segment in aligned_segments:
        word_segments += segment["words"]
    return {"segments": aligned_segments, "word_segments": word_segments}
