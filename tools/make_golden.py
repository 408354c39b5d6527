#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE's own code in this container.

Run from the repo root:  python tools/make_golden.py
Needs /root/reference (read-only).  Only data (inputs + expected outputs) is
written; no reference source travels.  What is imported:
  * /root/reference/whisperx/audio.py      as-is  (log_mel_spectrogram, pad_or_trim)
  * /root/reference/whisperx/utils.py      as-is  (the result writers, for option sets the gold files do not cover)
  * /root/reference/median_filter_fix.py   as-is  (median_filter_fixed: numpy + scipy)
  * /root/reference/whisperx/vads/vad.py   loaded by file path with one module stub (pyannote.core is not installed;
    Vad.merge_chunks touches neither it nor pandas) -- the package __init__ would pull in the pyannote pipeline
  * /root/reference/whisperx/batch_processor.py  with one module stub (mlx.core: Apple-only; none of create_chunks /
    create_batches / pad_batch / merge_results / _merge_overlapping_text touches it)
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
from whisperx_mlx_amd.synth import synth_audio  # noqa: E402


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


def make_median():
    """median_filter_fix.py:6-35 on seeded matrices of the shapes the DTW path feeds it ((N_tok, 1500) f32 after the
    softmax), plus the degenerate widths: rows shorter than / equal to the pad, ties, a 3-D input (the `else` branch)."""
    sys.path.insert(0, REF)
    import median_filter_fix as MF
    rng = np.random.default_rng(42)
    out = {}
    cases = {"soft_40x1500": None, "randn_7x64": rng.standard_normal((7, 64)).astype(np.float32),
             "ties_5x33": rng.integers(0, 4, (5, 33)).astype(np.float32),
             "short_3x3": rng.standard_normal((3, 3)).astype(np.float32),        # last dim <= pad: returned unchanged
             "min_2x4": rng.standard_normal((2, 4)).astype(np.float32),          # smallest row that is filtered
             "one_1x1500": rng.standard_normal((1, 1500)).astype(np.float32),
             "cube_2x3x50": rng.standard_normal((2, 3, 50)).astype(np.float32)}
    z = rng.standard_normal((40, 1500)).astype(np.float32) * 3.0
    e = np.exp(z - z.max(-1, keepdims=True))
    cases["soft_40x1500"] = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    for name, x in cases.items():
        for width in (7, 3):
            y = MF.median_filter_fixed(x.copy(), width)
            out[f"{name}_w{width}"] = np.asarray(y, dtype=np.float32)
        out[f"{name}_in"] = x
    np.savez_compressed(os.path.join(OUT, "median.npz"), **out)
    print("median fixtures:", len(cases), "inputs x 2 widths")


def _load_ref_file(modname, relpath, stubs):
    """one reference source file as a module, with `stubs` (name -> module) in sys.modules while it loads"""
    import importlib.util
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    try:
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def make_vad_merge():
    """Vad.merge_chunks (whisperx/vads/vad.py:20-53) on seeded lists of speech turns: the 81-window shape of a
    30-minute file, turns longer than chunk_size, back-to-back turns, a single turn, several chunk sizes."""
    pc = types.ModuleType("pyannote.core")
    pc.Annotation = type("Annotation", (), {})
    pc.Segment = type("Segment", (), {})
    pa = types.ModuleType("pyannote")
    pa.core = pc
    V = _load_ref_file("_ref_vad", "whisperx/vads/vad.py", {"pyannote": pa, "pyannote.core": pc})
    Turn = lambda s, e: types.SimpleNamespace(start=s, end=e, speaker="UNKNOWN")     # diarize.Segment's three fields
    rng = np.random.default_rng(7)
    cases = []

    def turns_random(n, mean_len, mean_gap, max_len):
        t, out = float(rng.uniform(0, 2)), []
        for _ in range(n):
            d = float(min(max_len, rng.exponential(mean_len) + 0.25))
            out.append((round(t, 3), round(t + d, 3)))
            t += d + float(rng.exponential(mean_gap))
        return out

    specs = [("speech_30", turns_random(400, 3.0, 0.6, 30.0), 30), ("speech_20", turns_random(150, 4.0, 0.3, 20.0), 20),
             ("long_turns", turns_random(40, 25.0, 1.0, 45.0), 30), ("single", [(1.5, 4.25)], 30),
             ("back_to_back", [(float(i), float(i + 1)) for i in range(95)], 30),
             ("exact_fit", [(0.0, 10.0), (10.0, 20.0), (20.0, 30.0), (30.0, 40.0), (40.0, 60.0), (60.5, 61.0)], 30),
             ("first_too_long", [(0.0, 31.0), (31.5, 33.0), (40.0, 75.0)], 30),
             ("tiny_chunks", turns_random(60, 1.0, 0.2, 4.0), 5)]
    for name, turns, cs in specs:
        merged = V.Vad.merge_chunks([Turn(s, e) for s, e in turns], cs, 0.5, 0.363)
        cases.append({"name": name, "chunk_size": cs, "turns": turns,
                      "merged": [{"start": m["start"], "end": m["end"], "segments": [list(x) for x in m["segments"]]} for m in merged]})
    with open(os.path.join(OUT, "vad_merge.json"), "w") as f:
        json.dump({"cases": cases}, f)
    print("vad merge fixtures:", [(c["name"], len(c["merged"])) for c in cases])


def make_batch_processor():
    """BatchProcessor.create_chunks / create_batches / pad_batch / merge_results (whisperx/batch_processor.py:47-148,
    186-276) on seeded segment lists and result texts; audio is a ramp so that every chunk's sample range is identified
    by its first and last sample."""
    mlx = types.ModuleType("mlx")
    mlx_core = types.ModuleType("mlx.core")
    mlx_core.array = type("array", (), {})          # named in two annotations at module level (:352)
    mlx.core = mlx_core
    BP = _load_ref_file("_ref_batch_processor", "whisperx/batch_processor.py", {"mlx": mlx, "mlx.core": mlx_core})
    rng = np.random.default_rng(11)
    words = "alpha bravo charlie delta echo foxtrot golf hotel india juliet kilo lima mike november oscar papa".split()
    cases = []
    specs = [("reference_test", 4, 30.0, 0.5, [(0.0, 10.0), (10.0, 50.0)], 60.0),
             ("ten_hours_head", 16, 30.0, 0.5, [(0.0, 3600.0)], 3600.0),
             ("mixed", 8, 30.0, 0.5, None, 900.0),
             ("short_chunks", 4, 20.0, 0.3, [(0.0, 19.99), (20.0, 40.0), (40.0, 40.001), (41.0, 105.37)], 110.0),
             ("exact_multiple", 3, 30.0, 0.5, [(0.0, 59.0), (59.0, 118.0), (120.0, 150.0)], 150.0),
             ("past_the_end", 2, 30.0, 0.5, [(50.0, 95.0)], 80.0)]
    for name, bs, dur, ov, segs, total in specs:
        if segs is None:
            t, segs = 0.0, []
            while t < total - 1:
                d = float(rng.choice([2.5, 11.0, 29.9, 30.0, 30.1, 64.2, 95.0]))
                segs.append((round(t, 3), round(min(t + d, total), 3)))
                t += d + float(rng.uniform(0, 1.5))
        audio = np.arange(int(total * 16000), dtype=np.float32)
        p = BP.BatchProcessor(batch_size=bs, chunk_duration=dur, overlap=ov)
        seg_dicts = [{"start": a, "end": b} for a, b in segs]
        chunks = p.create_chunks(audio, seg_dicts)
        batches = p.create_batches(chunks)
        pads = []
        for b in batches[:4]:
            if all(len(c.audio) for c in b):
                pa_, lens = p.pad_batch(b)
                pads.append({"shape": list(pa_.shape), "lengths": [int(x) for x in lens],
                             "row_sums": [float(r.astype(np.float64).sum()) for r in pa_]})
            else:
                pads.append(None)
        results = []
        for k, c in enumerate(chunks):
            n = int(rng.integers(0, 14))
            txt = " ".join(words[int(j)] for j in rng.integers(0, len(words), n))
            results.append({"text": ("  " if k % 3 == 0 else "") + txt + (" " if k % 2 else "")})
        drop = set(int(i) for i in rng.choice(len(chunks), size=min(2, len(chunks) // 4), replace=False)) if len(chunks) > 8 else set()
        kept = [i for i in range(len(chunks)) if i not in drop]
        order = [int(i) for i in rng.permutation(kept)]                      # results arrive out of order, some missing
        merged = p.merge_results([chunks[i] for i in order], [results[i] for i in order], seg_dicts)
        cases.append({"name": name, "batch_size": bs, "chunk_duration": dur, "overlap": ov, "total_s": total, "segments": segs,
                      "chunks": [{"start": c.start_time, "end": c.end_time, "segment_idx": c.segment_idx, "n": int(len(c.audio)),
                                  "first": float(c.audio[0]) if len(c.audio) else None,
                                  "last": float(c.audio[-1]) if len(c.audio) else None} for c in chunks],
                      "batch_lens": [len(b) for b in batches], "pads": pads, "results": results, "order": order, "merged": merged})
    with open(os.path.join(OUT, "batch_processor.json"), "w") as f:
        json.dump({"cases": cases}, f)
    print("batch_processor fixtures:", [(c["name"], len(c["chunks"])) for c in cases])


def _jsonable(o):
    if isinstance(o, (np.floating,)):
        return None if np.isnan(o) else float(o)
    if isinstance(o, (np.integer,)):
        return int(o)
    raise TypeError(type(o))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])
    if only:                    # e.g. `python tools/make_golden.py median vad batch`: the round-3 fixtures alone
        if "median" in only:
            make_median()
        if "vad" in only:
            make_vad_merge()
        if "batch" in only:
            make_batch_processor()
        sys.exit(0)
    make_logmel()
    make_median()
    make_vad_merge()
    make_batch_processor()
    install_stubs()
    make_ctc()
    make_align()
    make_writers()
