#!/usr/bin/env python3
"""Builds tests/golden/gold30m_windows.json from the reference's result artefact /root/reference/30m.json
(743 segments of a whisper-large-v3 run, each with its token ids).

The artefact stores, per result segment, the slice of the decode window's sampled tokens that produced it
(<|t0|> text... <|t1|>), with absolute start times.  Segments whose window start (segment start minus the leading
timestamp token's offset) agrees belong to one 30 s decode window; concatenating their token lists restores the
window's sampled sequence.  The fixture is data only: window start (s) and the token ids.

    python tools/make_gold_tokens.py            # needs /root/reference (this container only)
"""
import json
import os

REF = "/root/reference/30m.json"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "gold30m_windows.json")
TIMESTAMP_BEGIN = 50365          # <|0.00|> of the large-v3 vocabulary: every window in the artefact starts with it or just after


def main():
    segs = json.load(open(REF))["segments"]
    windows, cur = [], None
    for s in segs:
        t = s["tokens"]
        assert t[0] >= TIMESTAMP_BEGIN, s["id"]
        w0 = round(s["start"] - (t[0] - TIMESTAMP_BEGIN) * 0.02, 3)
        if cur is None or abs(cur["start"] - w0) > 0.011:
            cur = {"start": w0, "tokens": [], "avg_logprob": s["avg_logprob"], "no_speech_prob": s["no_speech_prob"],
                   "compression_ratio": s["compression_ratio"], "first_segment": s["id"], "text": ""}
            windows.append(cur)
        cur["tokens"] += t
        cur["text"] += s["text"]
    out = {"source": "reference 30m.json (whisper-large-v3, 743 segments)", "timestamp_begin": TIMESTAMP_BEGIN,
           "n_segments": len(segs), "windows": windows}
    with open(OUT, "w") as f:
        json.dump(out, f, separators=(",", ":"), ensure_ascii=False)
    n = sum(len(w["tokens"]) for w in windows)
    print(f"{len(windows)} windows, {n} tokens ({n / 60:.1f} per 30 s of the 30 min file) -> {OUT}")
    # the 743 segment texts of the same run: input of the sentence splitter the reference applies before alignment
    # (alignment.py:191-194); its output survives in the gold standard's 779 aligned segments
    texts = os.path.join(os.path.dirname(OUT), "gold30m_segment_texts.json")
    with open(texts, "w") as f:
        json.dump({"source": "reference 30m.json segment texts (whisper-large-v3 run)", "texts": [s["text"] for s in segs]}, f,
                  ensure_ascii=False, separators=(",", ":"))
    print(f"{len(segs)} segment texts -> {texts}")


if __name__ == "__main__":
    main()
