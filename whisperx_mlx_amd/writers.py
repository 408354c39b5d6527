"""Result writers: txt / vtt / srt / tsv / json / aud.

Host-side text formatting, outside the HIP hot path; same observable behaviour as the
reference's `whisperx/utils.py:170-440` (`format_timestamp`, `WriteTXT`, `SubtitlesWriter`,
`WriteVTT`, `WriteSRT`, `WriteTSV`, `WriteAudacity`, `WriteJSON`, `get_writer`): given the same
result dict the files are byte-identical to `whisperx-large-v3-gold-standard/30m.{txt,vtt,srt,tsv}`
(tests/test_writers.py pins that on the gold JSON).  Own implementation: cues are built by a
small state machine (`_Cues`) instead of nested generators.
"""
import json
import os
import re
from typing import Callable, Dict, Iterator, List, Optional, TextIO, Tuple

LANGUAGES_WITHOUT_SPACES = ("ja", "zh")      # utils.py:107


def format_timestamp(seconds: float, always_include_hours: bool = False, decimal_marker: str = ".") -> str:
    """utils.py:170-189: [HH:]MM:SS<marker>mmm, rounded to the millisecond."""
    if seconds < 0:
        raise AssertionError("non-negative timestamp expected")
    total_ms = round(seconds * 1000.0)
    hours, rem = divmod(total_ms, 3_600_000)
    minutes, rem = divmod(rem, 60_000)
    secs, ms = divmod(rem, 1_000)
    head = f"{hours:02d}:" if (always_include_hours or hours > 0) else ""
    return f"{head}{minutes:02d}:{secs:02d}{decimal_marker}{ms:03d}"


class ResultWriter:
    extension = ""

    def __init__(self, output_dir: str):
        self.output_dir = output_dir

    def __call__(self, result: dict, audio_path: str, options: Optional[dict] = None):
        stem = os.path.splitext(os.path.basename(audio_path))[0]
        path = os.path.join(self.output_dir, f"{stem}.{self.extension}")
        with open(path, "w", encoding="utf-8") as f:
            self.write_result(result, file=f, options=options or {})
        return path

    def write_result(self, result: dict, file: TextIO, options: dict):
        raise NotImplementedError


class WriteTXT(ResultWriter):
    extension = "txt"

    def write_result(self, result, file, options):
        for seg in result["segments"]:
            text = seg["text"].strip()
            spk = seg.get("speaker")
            file.write(f"[{spk}]: {text}\n" if spk is not None else f"{text}\n")
            file.flush()


class _Cues:
    """Groups aligned words into subtitle cues (utils.py:229-283).  A cue is a list of word dicts
    (their "word" possibly prefixed by a line break) plus the (start, end, speaker) of the segment
    every word came from."""

    def __init__(self, max_line_width: Optional[int], max_line_count: Optional[int]):
        self.width = 1000 if max_line_width is None else max_line_width
        self.max_lines = max_line_count
        # without both limits the segmentation of the input is kept: one cue per segment
        self.keep_segments = max_line_count is None or max_line_width is None

    def build(self, segments: List[dict]) -> Iterator[Tuple[List[dict], List[tuple]]]:
        cue: List[dict] = []
        spans: List[tuple] = []
        line_len, n_lines = 0, 1
        last_start = segments[0]["start"]
        for seg in segments:
            for idx, src in enumerate(seg["words"]):
                w = dict(src)
                pause = (not self.keep_segments) and ("start" in w) and (w["start"] - last_start > 3.0)
                fits = line_len + len(w["word"]) <= self.width
                new_segment = idx == 0 and bool(cue) and self.keep_segments
                if line_len > 0 and fits and not pause and not new_segment:
                    line_len += len(w["word"])                      # same line
                else:
                    w["word"] = w["word"].strip()
                    if (cue and self.max_lines is not None and (pause or n_lines >= self.max_lines)) or new_segment:
                        yield cue, spans                            # cue break
                        cue, spans, n_lines = [], [], 1
                    elif line_len > 0:
                        n_lines += 1                                # line break inside the cue
                        w["word"] = "\n" + w["word"]
                    line_len = len(w["word"].strip())
                cue.append(w)
                spans.append((seg["start"], seg["end"], seg.get("speaker")))
                if "start" in w:
                    last_start = w["start"]
        if cue:
            yield cue, spans


class SubtitlesWriter(ResultWriter):
    always_include_hours = False
    decimal_marker = "."

    def _ts(self, seconds: float) -> str:
        return format_timestamp(seconds, self.always_include_hours, self.decimal_marker)

    def iterate_result(self, result: dict, options: dict) -> Iterator[Tuple[str, str, str]]:
        segments = result["segments"]
        if not segments:
            return
        if "words" not in segments[0]:                               # utils.py:320-327: unaligned result
            for seg in segments:
                text = seg["text"].strip().replace("-->", "->")
                if "speaker" in seg:
                    text = f"[{seg['speaker']}]: {text}"
                yield self._ts(seg["start"]), self._ts(seg["end"]), text
            return
        highlight = bool(options.get("highlight_words", False))
        joiner = "" if result.get("language") in LANGUAGES_WITHOUT_SPACES else " "
        cues = _Cues(options.get("max_line_width"), options.get("max_line_count"))
        for cue, spans in cues.build(segments):
            seg_start, seg_end, speaker = spans[0]
            t0, t1 = self._ts(seg_start), self._ts(seg_end)
            text = joiner.join(w["word"] for w in cue)
            prefix = f"[{speaker}]: " if speaker is not None else ""
            if not (highlight and any("start" in w for w in cue)):
                yield t0, t1, prefix + text
                continue
            # one cue per word with that word underlined; gaps show the plain text (utils.py:297-317)
            words = [w["word"] for w in cue]
            cursor = t0
            for i, w in enumerate(cue):
                if "start" not in w:
                    continue
                ws, we = self._ts(w["start"]), self._ts(w["end"])
                if cursor != ws:
                    yield cursor, ws, prefix + text
                marked = [re.sub(r"^(\s*)(.*)$", r"\1<u>\2</u>", x) if j == i else x for j, x in enumerate(words)]
                yield ws, we, prefix + " ".join(marked)
                cursor = we


class WriteVTT(SubtitlesWriter):
    extension = "vtt"
    always_include_hours = False
    decimal_marker = "."

    def write_result(self, result, file, options):
        file.write("WEBVTT\n\n")
        for start, end, text in self.iterate_result(result, options):
            file.write(f"{start} --> {end}\n{text}\n\n")
            file.flush()


class WriteSRT(SubtitlesWriter):
    extension = "srt"
    always_include_hours = True
    decimal_marker = ","

    def write_result(self, result, file, options):
        for n, (start, end, text) in enumerate(self.iterate_result(result, options), start=1):
            file.write(f"{n}\n{start} --> {end}\n{text}\n\n")
            file.flush()


class WriteTSV(ResultWriter):
    """start / end in integer milliseconds, tab separated (utils.py:362-378)."""
    extension = "tsv"

    def write_result(self, result, file, options):
        file.write("start\tend\ttext\n")
        for seg in result["segments"]:
            text = seg["text"].strip().replace("\t", " ")
            file.write(f"{round(1000 * seg['start'])}\t{round(1000 * seg['end'])}\t{text}\n")
            file.flush()


class WriteAudacity(ResultWriter):
    """Audacity label track: seconds, tab separated, speaker in [[..]] (utils.py:381-400)."""
    extension = "aud"

    def write_result(self, result, file, options):
        for seg in result["segments"]:
            spk = f"[[{seg['speaker']}]]" if "speaker" in seg else ""
            text = seg["text"].strip().replace("\t", " ")
            file.write(f"{seg['start']}\t{seg['end']}\t{spk}{text}\n")
            file.flush()


class WriteJSON(ResultWriter):
    extension = "json"

    def write_result(self, result, file, options):
        json.dump(result, file, ensure_ascii=False)


_WRITERS: Dict[str, type] = {"txt": WriteTXT, "vtt": WriteVTT, "srt": WriteSRT, "tsv": WriteTSV, "json": WriteJSON}
_OPTIONAL: Dict[str, type] = {"aud": WriteAudacity}


def get_writer(output_format: str, output_dir: str) -> Callable[[dict, str, dict], None]:
    """utils.py:411-436: "all" writes txt, vtt, srt, tsv and json."""
    if output_format == "all":
        ws = [cls(output_dir) for cls in _WRITERS.values()]

        def write_all(result: dict, audio_path: str, options: Optional[dict] = None):
            for w in ws:
                w(result, audio_path, options)

        return write_all
    if output_format in _OPTIONAL:
        return _OPTIONAL[output_format](output_dir)
    if output_format not in _WRITERS:
        raise ValueError(f"unknown output format {output_format!r}")
    return _WRITERS[output_format](output_dir)
