#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE's own code in this container.

Run from the repo root:  python tools/make_golden.py
Needs /root/reference (read-only).  Only data (inputs + expected outputs) is
written; no reference source travels.  What is imported:
  * /root/reference/whisperx/audio.py      as-is  (log_mel_spectrogram, pad_or_trim)
  * /root/reference/whisperx/utils.py      as-is  (the result writers, for option sets the gold files do not cover)
  * /root/reference/whisperx/alignment.py  with two module stubs (torchaudio,
    nltk.tokenize.punkt are not installed): the punkt stub splits sentences with
    the simple rule in `simple_spans` below and the spans are saved in the
    fixture, so the product's assembly logic is checked on multi-sentence text.
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)


def simple_spans(text):
    """Deterministic stand-in for PunktSentenceTokenizer.span_tokenize: break after
    [.?!] followed by whitespace; spans exclude the separating whitespace."""
    spans, start, i, n = [], 0, 0, len(text)
    while i < n:
        if text[i] in ".?!" and i + 1 < n and text[i + 1].isspace():
            spans.append((start, i + 1))
            j = i + 1
            while j < n and text[j].isspace():
                j += 1
            start = j
            i = j
        else:
            i += 1
    if start < n:
        spans.append((start, n))
    return spans


def install_stubs():
    import transformers  # noqa: F401  (real one must be imported first)
    ta = types.ModuleType("torchaudio")
    ta.pipelines = types.SimpleNamespace(__all__=[], __dict__={})
    sys.modules["torchaudio"] = ta
    nltk = types.ModuleType("nltk")
    tok = types.ModuleType("nltk.tokenize")
    punkt = types.ModuleType("nltk.tokenize.punkt")

    class PunktParameters:
        abbrev_types = set()

    class PunktSentenceTokenizer:
        def __init__(self, params=None):
            pass

        def span_tokenize(self, text):
            return simple_spans(text)

    punkt.PunktParameters = PunktParameters
    punkt.PunktSentenceTokenizer = PunktSentenceTokenizer
    sys.modules["nltk"] = nltk
    sys.modules["nltk.tokenize"] = tok
    sys.modules["nltk.tokenize.punkt"] = punkt


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.synth import synth_audio  # noqa: E402


def make_logmel():
    from whisperx import audio as A
    sample = np.load(os.path.join(REF, "audio_sample.npy"))
    pcm16 = np.round(sample * 32768.0).astype(np.int16)
    assert np.array_equal(pcm16.astype(np.float32) / 32768.0, sample)
    out = {"audio_sample_i16": pcm16}
    # (1) whole-clip mel, both filterbanks (audio.py:112-159 as called at mlx_lightning.py:163)
    for n_mels in (80, 128):
        m = A.log_mel_spectrogram(sample, n_mels).numpy()
        out[f"sample_mel{n_mels}"] = m.astype(np.float32)
    # (2) chunk form: pad_or_trim to 30 s then mel (path C); keep slices + checksums
    for n_mels in (80, 128):
        x = A.pad_or_trim(sample, A.N_SAMPLES)
        m = A.log_mel_spectrogram(x, n_mels).numpy()
        assert m.shape == (n_mels, 3000)
        out[f"chunk_mel{n_mels}_head"] = m[:, :64].copy()
        out[f"chunk_mel{n_mels}_mid"] = m[:, 468:532].copy()     # around the 5 s speech/zero edge
        out[f"chunk_mel{n_mels}_tail"] = m[:, -64:].copy()
        out[f"chunk_mel{n_mels}_stats"] = np.array(
            [m.mean(dtype=np.float64), m.min(), m.max(), (m.astype(np.float64) ** 2).sum()])
    # (3) seeded synthetic, ragged lengths incl. tiny and exact-30 s
    for seed, n in ((1, 16000), (2, 47999), (3, 480000), (4, 1000)):
        x = A.pad_or_trim(synth_audio(seed, n), A.N_SAMPLES)
        m = A.log_mel_spectrogram(x, 128).numpy()
        out[f"synth{seed}_n"] = np.array([n])
        out[f"synth{seed}_head"] = m[:, :48].copy()
        k = n // 160
        lo = max(0, min(k - 24, 3000 - 48))
        out[f"synth{seed}_edge_lo"] = np.array([lo])
        out[f"synth{seed}_edge"] = m[:, lo:lo + 48].copy()
        out[f"synth{seed}_tail"] = m[:, -48:].copy()
        out[f"synth{seed}_stats"] = np.array(
            [m.mean(dtype=np.float64), m.min(), m.max(), (m.astype(np.float64) ** 2).sum()])
    np.savez_compressed(os.path.join(OUT, "logmel.npz"), **out)
    # the reference's filterbank asset (data) -> checks our own Slaney-mel generator
    with np.load(os.path.join(REF, "whisperx", "assets", "mel_filters.npz")) as f:
        np.savez_compressed(os.path.join(OUT, "mel_filters_ref.npz"),
                            mel_80=f["mel_80"], mel_128=f["mel_128"])
    print("logmel fixtures:", sorted(out)[:6], "...")


def make_ctc():
    from whisperx import alignment as AL
    cases = {}
    specs = [
        # name, seed, T, V, tokens
        ("wild", 0, 149, 32, [5, 9, 4, -1, 7, 7, 12]),
        ("single", 1, 40, 29, [3]),
        ("two", 2, 12, 29, [3, 8]),
        ("tight", 3, 9, 29, [1, 2, 3, 4, 5, 6, 7, 8]),          # N == T-1
        ("toolong", 4, 6, 29, [1, 2, 3, 4, 5, 6, 7, 8, 9]),       # N > T -> backtrack fails
        ("long", 5, 700, 32, None),
        ("blank5", 6, 200, 32, [4, 4, -1, -1, 9, 10, 4, 31, 1]),
    ]
    for name, seed, T, V, tokens in specs:
        g = torch.Generator().manual_seed(seed)
        blank = 5 if name == "blank5" else 0
        em = torch.log_softmax(torch.randn(T, V, generator=g) * 2.0, dim=-1)
        if tokens is None:
            tokens = torch.randint(1, V, (130,), generator=g).tolist()
            for k in range(0, 130, 17):
                tokens[k] = -1
        trellis = AL.get_trellis(em, tokens, blank)
        path = AL.backtrack_beam(trellis, em, tokens, blank, beam_width=2)
        cases[f"{name}_emission"] = em.numpy()
        cases[f"{name}_tokens"] = np.array(tokens, dtype=np.int32)
        cases[f"{name}_blank"] = np.array([blank], dtype=np.int32)
        cases[f"{name}_trellis"] = trellis.numpy()
        if path is None:
            cases[f"{name}_ok"] = np.array([0], dtype=np.int32)
        else:
            cases[f"{name}_ok"] = np.array([1], dtype=np.int32)
            cases[f"{name}_path_tok"] = np.array([p.token_index for p in path], dtype=np.int32)
            cases[f"{name}_path_time"] = np.array([p.time_index for p in path], dtype=np.int32)
            cases[f"{name}_path_score"] = np.array([p.score for p in path], dtype=np.float64)
            text = "".join(chr(97 + (k % 26)) for k in range(len(tokens)))
            segs = AL.merge_repeats(path, text)
            cases[f"{name}_seg_start"] = np.array([s.start for s in segs], dtype=np.int32)
            cases[f"{name}_seg_end"] = np.array([s.end for s in segs], dtype=np.int32)
            cases[f"{name}_seg_score"] = np.array([s.score for s in segs], dtype=np.float64)
        # beam width 5 as well (signature default, alignment.py:500)
        p5 = AL.backtrack_beam(trellis, em, tokens, blank, beam_width=5)
        if p5 is not None:
            cases[f"{name}_path5_tok"] = np.array([p.token_index for p in p5], dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "ctc.npz"), **cases)
    print("ctc fixtures:", len(cases), "arrays")


class FakeCTCModel(torch.nn.Module):
    """Deterministic stand-in for Wav2Vec2ForCTC so that align() can run end to end
    without weights: frames of 320 samples (400 window) -> seeded linear -> logits,
    sharpened so the alignment is not degenerate.  Emissions are saved in the
    fixture, so the product test injects exactly these."""

    def __init__(self, vocab, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.w = torch.randn(400, vocab, generator=g) * 4.0
        self.calls = []

    def forward(self, wave):
        n = wave.shape[-1]
        T = (n - 400) // 320 + 1
        idx = torch.arange(400)[None, :] + 320 * torch.arange(T)[:, None]
        fr = wave[0][idx]
        logits = (fr @ self.w)[None]
        self.calls.append(logits[0].clone())
        return types.SimpleNamespace(logits=logits)


def make_align():
    from whisperx import alignment as AL
    sample = np.load(os.path.join(REF, "audio_sample.npy"))
    labels = ["<pad>", "<s>", "</s>", "<unk>", "|"] + list("etaonihsrdlumwcfgypbvk'xjqz")
    dictionary = {c.lower(): i for i, c in enumerate(labels)}
    meta = {"language": "en", "dictionary": dictionary, "type": "huggingface"}
    docs = {}
    scenarios = {
        "short_json": [   # the two real segments of short.json that audio_sample.npy covers
            {"start": 0.976, "end": 2.539, "text": "That's why he's so fucking famous, bro."},
            {"start": 3.681, "end": 5.0, "text": "That's why Gordon Ramsey's so famous."},
        ],
        "edge_cases": [
            {"start": 0.0, "end": 1.2, "text": " Hello there.  General Kenobi! You are 2 bold. "},
            {"start": 1.2, "end": 1.21, "text": "tiny"},                 # < 400 samples -> padded
            {"start": 1.5, "end": 2.5, "text": "12345 67"},               # only wildcards
            {"start": 2.5, "end": 3.0, "text": "€ → ∑"},                  # wildcards + spaces
            {"start": 7.0, "end": 8.0, "text": "beyond the audio"},       # start >= duration
            {"start": 3.0, "end": 3.1, "text": "this text is far too long for three frames"},
            {"start": 3.2, "end": 4.9, "text": "Mr. Smith went. He came back? Yes."},
            {"start": 4.0, "end": 4.5, "text": "   "},                    # nothing alignable... spaces only
        ],
    }
    for name, segs in scenarios.items():
        model = FakeCTCModel(len(labels), seed=7)
        for rc in (False, True):
            model.calls.clear()
            res = AL.align([dict(s) for s in segs], model, meta, sample, "cpu",
                           return_char_alignments=rc)
            key = name + ("_chars" if rc else "")
            docs[key] = {
                "segments_in": segs,
                "sentence_spans": [simple_spans(s["text"]) for s in segs],
                "result": json.loads(json.dumps(res, default=_jsonable)),
            }
        np.savez_compressed(os.path.join(OUT, f"align_{name}_emissions.npz"),
                            **{f"call{i}": torch.log_softmax(c, -1).numpy() for i, c in enumerate(model.calls)})
    docs["dictionary"] = dictionary
    with open(os.path.join(OUT, "align.json"), "w") as f:
        json.dump(docs, f, indent=1, ensure_ascii=False)
    print("align fixtures:", list(docs))


def make_writers():
    """The reference's own SubtitlesWriter / WriteTSV / WriteTXT / WriteAudacity (whisperx/utils.py) on the first
    60 segments of its gold result dict plus three hand-made segments (speaker label, a word without timing,
    a language written without spaces), for option sets the gold files do not cover."""
    import io
    import whisperx.utils as U
    gold = json.load(open(os.path.join(REF, "whisperx-large-v3-gold-standard", "30m.json"), encoding="utf-8"))
    res = {"language": "en", "segments": gold["segments"][:60]}
    res["segments"][3] = dict(res["segments"][3], speaker="SPEAKER_01")
    w5 = [dict(w) for w in res["segments"][5]["words"]]
    for k in ("start", "end", "score"):
        w5[1].pop(k, None)
    res["segments"][5] = dict(res["segments"][5], words=w5)
    ja = {"language": "ja", "segments": [{"start": 0.5, "end": 2.0, "text": "今日は晴れ", "words": [
        {"word": "今日", "start": 0.5, "end": 0.9}, {"word": "は", "start": 0.9, "end": 1.0}, {"word": "晴れ", "start": 1.2, "end": 2.0}]}]}
    cases = []
    for name, result in (("en60", res), ("ja", ja)):
        for mlw, mlc, hl in ((42, 2, False), (None, None, True), (30, 1, True), (20, 3, False), (42, None, False)):
            opt = {"max_line_width": mlw, "max_line_count": mlc, "highlight_words": hl}
            out = {}
            for ext, cls in (("srt", U.WriteSRT), ("vtt", U.WriteVTT), ("tsv", U.WriteTSV), ("txt", U.WriteTXT), ("aud", U.WriteAudacity)):
                buf = io.StringIO()
                cls(".").write_result(result, file=buf, options=opt)
                out[ext] = buf.getvalue()
            cases.append({"result": name, "options": opt, "out": out})
    with open(os.path.join(OUT, "writers_opts.json"), "w", encoding="utf-8") as f:
        json.dump({"results": {"en60": res, "ja": ja}, "cases": cases}, f, ensure_ascii=False)
    print("writer fixtures:", len(cases), "cases")


def _jsonable(o):
    if isinstance(o, (np.floating,)):
        return None if np.isnan(o) else float(o)
    if isinstance(o, (np.integer,)):
        return int(o)
    raise TypeError(type(o))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    make_logmel()
    install_stubs()
    make_ctc()
    make_align()
    make_writers()
