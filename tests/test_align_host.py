"""CPU: align() host logic (char cleaning, timestamps, word/sentence assembly, NaN
interpolation, failure branches) against the result dicts the REFERENCE's align()
produced for the same emissions (tests/golden/align.json, tools/make_golden.py).
The numeric backend is injected: fixture emissions + the oracle's CTC DP (the GPU path
is covered by -m gpu tests)."""
import json
import math
import os

import numpy as np
import pytest

from oracle import ctc as OC
from tests.conftest import GOLDEN
from whisperx_mlx_amd import alignment as AL


def _load():
    with open(os.path.join(GOLDEN, "align.json")) as f:
        return json.load(f)


def _norm(o):
    """NaN -> None so that dict equality works (the fixture stored NaN as null)."""
    if isinstance(o, float):
        return None if math.isnan(o) else o
    if isinstance(o, dict):
        return {k: _norm(v) for k, v in o.items()}
    if isinstance(o, list):
        return [_norm(v) for v in o]
    return o


def _oracle_aligner(emissions):
    calls = iter(emissions)

    def run(waveforms, token_lists, blank_id, beam):
        out = []
        for wav, toks in zip(waveforms, token_lists):
            em = next(calls)
            tr = OC.get_trellis(em, toks, blank_id)
            path = OC.backtrack_beam(tr, em, toks, blank_id, beam)
            if path is None:
                out.append((em.shape[0], None, None))
            else:
                out.append((em.shape[0], [p[0] for p in path], [p[2] for p in path]))
        return out
    return run


@pytest.mark.parametrize("name", ["short_json", "edge_cases"])
@pytest.mark.parametrize("chars", [False, True])
def test_align_matches_reference_result(name, chars):
    docs = _load()
    doc = docs[name + ("_chars" if chars else "")]
    em = np.load(os.path.join(GOLDEN, f"align_{name}_emissions.npz"))
    emissions = [em[f"call{i}"] for i in range(len(em.files))]
    audio = np.load(os.path.join(GOLDEN, "logmel.npz"))["audio_sample_i16"].astype(np.float32) / 32768.0
    meta = {"language": "en", "dictionary": docs["dictionary"], "type": "hip"}
    spans = doc["sentence_spans"]
    res = AL.align([dict(s) for s in doc["segments_in"]], None, meta, audio, "cpu", return_char_alignments=chars,
                   _aligner=_oracle_aligner(emissions), _sentence_spans=lambda sdx, text: [tuple(x) for x in spans[sdx]])
    assert _norm(res) == _norm(doc["result"])


def test_interpolate_nans_nearest():
    nan = float("nan")
    assert AL.interpolate_nans([nan, 1.0, nan, nan, 4.0, nan]) == [1.0, 1.0, 1.0, 4.0, 4.0, 4.0]
    assert AL.interpolate_nans([nan, 2.0, nan]) == [2.0, 2.0, 2.0]
    out = AL.interpolate_nans([nan, nan])
    assert all(math.isnan(v) for v in out)


def test_sentence_spans_rules():
    t = "Mr. Smith went. He came back? Yes."
    assert [t[a:b] for a, b in AL.sentence_spans(t)] == ["Mr. Smith went.", "He came back?", "Yes."]
    assert AL.sentence_spans("no punctuation here") == [(0, 19)]
    t2 = "It cost 3.5 dollars. J. Doe paid... later. Dr. Who?"
    assert [t2[a:b] for a, b in AL.sentence_spans(t2)] == ["It cost 3.5 dollars.", "J. Doe paid... later.", "Dr. Who?"]


def test_sentence_splitter_against_the_reference_run():
    """`sentence_spans` (our stand-in for the untrained PunktSentenceTokenizer of alignment.py:191-194; nltk is not a
    dependency) on REAL text the reference processed: the 743 Whisper segments of /root/reference/30m.json split into
    sentences must reproduce the segment texts of the reference's aligned gold standard (779 segments: the same audio,
    split by nltk).  The two artefacts come from two runs of the reference, so a handful of segments differ in their
    WORDS; a difference in where a text is cut, with the words equal, would be a splitter bug and fails here."""
    import difflib
    import gzip
    import json
    import os
    from whisperx_mlx_amd.alignment import sentence_spans
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    texts = json.load(open(os.path.join(g, "gold30m_segment_texts.json")))["texts"]
    gold = [s["text"].strip() for s in json.load(gzip.open(os.path.join(g, "gold30m", "30m.json.gz"), "rt"))["segments"]]
    mine, made_by_splitter = [], []          # made_by_splitter[k]: the cut BEFORE sentence k lies inside one source segment
    for t in texts:
        for k, (a, b) in enumerate(sentence_spans(t)):
            mine.append(t[a:b].strip())
            made_by_splitter.append(k > 0)
    assert len(texts) == 743 and len(gold) == 779
    assert sum(made_by_splitter) >= 30                               # real cuts are exercised
    sm = difflib.SequenceMatcher(a=mine, b=gold, autojunk=False)
    assert sm.ratio() > 0.98
    for tag, i1, i2, j1, j2 in sm.get_opcodes():
        if tag == "equal":
            continue
        a, b = " ".join(mine[i1:i2]).split(), " ".join(gold[j1:j2]).split()
        if a != b:
            continue          # the two runs transcribed these words differently: nothing to learn about the splitter
        # same words: every cut the SPLITTER made (not the ones between Whisper segments, which differ between the runs)
        # must be a cut of the reference as well
        def cuts(sents):
            out, n = [], 0
            for x in sents[:-1]:
                n += len(x.split())
                out.append(n)
            return out
        ref_cuts = set(cuts(gold[j1:j2]))
        for k, c in enumerate(cuts(mine[i1:i2])):
            assert not made_by_splitter[i1 + k + 1] or c in ref_cuts, (mine[i1:i2], gold[j1:j2])


def test_linear_assembly_equals_the_row_filter_statement():
    """the shipped char -> word -> sentence assembly (columns as lists, a word = a contiguous run of characters; linear in
    the text) against the per-row filtering it replaced (tests/align_loop_reference.py, the reference's pandas logic of
    alignment.py:296-343 restated row by row): identical result dicts -- every time, score and key -- on random transcripts
    with leading / trailing / double spaces, characters outside the dictionary, several sentences per segment, segments
    that fail, with and without character alignments."""
    from tests import align_loop_reference as OLD
    rng = np.random.default_rng(17)
    meta = {"language": "en", "dictionary": _load()["dictionary"], "type": "hip"}
    alphabet = list("abcdefghijklmnopqrstuvwxyz'") + ["é", "3", "-"]

    def random_text():
        words = []
        for _ in range(int(rng.integers(1, 40))):
            w = "".join(rng.choice(alphabet) for _ in range(int(rng.integers(1, 14))))
            words.append(w + str(rng.choice(["", "", "", ".", "?", ",", "..."])))
        t = " ".join(words)
        if rng.random() < 0.3:
            t = t.replace(" ", "  ", 1)
        return str(rng.choice(["", " ", "  "])) + t + str(rng.choice(["", " "]))

    def aligner(waveforms, token_lists, blank_id, beam):
        out = []
        for wav, toks in zip(waveforms, token_lists):
            T, N = max(0, (len(wav) - 400) // 320 + 1), len(toks)
            if T < N or T < 2 or rng.random() < 0.05:
                out.append((T, None, None))
                continue
            cuts = np.sort(rng.choice(np.arange(1, T), size=N - 1, replace=False)) if N > 1 else np.array([], dtype=int)
            path = np.searchsorted(cuts, np.arange(T), side="right")
            out.append((T, path.tolist(), rng.random(T).astype(np.float32).astype(np.float64).tolist()))
        return out

    for trial in range(12):
        items = []
        for _ in range(int(rng.integers(1, 5))):
            n = int(rng.integers(32000, 300000))
            segs, t = [], 0.0
            for _ in range(int(rng.integers(1, 4))):
                d = float(rng.uniform(0.5, n / 16000.0 / 3))
                segs.append({"start": round(t, 2), "end": round(t + d, 2), "text": random_text()})
                t += d
            if rng.random() < 0.2:
                segs.append({"start": n / 16000.0 + 1.0, "end": n / 16000.0 + 2.0, "text": "beyond the audio"})
            items.append((segs, np.zeros(n, dtype=np.float32)))
        chars = bool(trial % 2)
        state = rng.bit_generator.state
        new = AL.align_batch([([dict(s) for s in segs], a) for segs, a in items], None, meta, "cpu", return_char_alignments=chars,
                             _aligner=aligner)
        rng.bit_generator.state = state                   # the same random paths for the second run
        old = OLD.align_batch([([dict(s) for s in segs], a) for segs, a in items], None, meta, "cpu", return_char_alignments=chars,
                              _aligner=aligner)
        assert _norm(new) == _norm(old), trial


def test_round3_is_pythons_round():
    """alignment._round3 (vectorised) against round(v, 3) on random values, values a hair from the rounding boundary (where
    x * 1000 rounds across the half), negative values, and the values a path really produces (frame * ratio + t1)"""
    rng = np.random.default_rng(5)
    xs = [rng.random(20000) * 40.0, rng.random(2000) - 0.5, np.array([0.0, 0.0005, 0.0015, 2.675, 1.0005, 1e-9, 29.9995, 12.3455])]
    k = rng.integers(0, 40000, 20000).astype(np.float64)
    for eps in (0.0, 1e-17, -1e-17, 3e-16, -3e-16, 1e-13, -1e-13):
        xs.append((k + 0.5) / 1000.0 + eps)
        xs.append(np.nextafter((k + 0.5) / 1000.0, np.inf if eps >= 0 else -np.inf))
    ratio, t1 = 27.34 / 1366, 3.07
    xs.append(np.arange(1367, dtype=np.float64) * ratio + t1)
    for x in xs:
        assert AL._round3(x) == [round(float(v), 3) for v in x]


def test_align_batch_assembles_as_forwards_arrive():
    """align_batch takes the aligner's results forward by forward (`batches`, round 5: the GPU runs the next forward while
    the host assembles the words of the last) and assembles a pair as soon as the last of its segments is back: whatever
    the cut and the order of arrival, the dicts and traces are those of the all-at-once call"""
    import numpy as np
    from whisperx_mlx_amd import alignment as AL
    with open(os.path.join(GOLDEN, "align.json")) as f:
        dictionary = json.load(f)["dictionary"]
    meta = {"language": "en", "dictionary": dictionary, "type": "hip"}
    rng = np.random.default_rng(4)
    words = ["that's", "what", "they", "said", "Mr.", "Smith", "went", "home.", "He", "came", "back?", "Yes.", "naïve", "3.5", "dollars"]
    items = []
    for i in range(11):
        segs, t = [], 0.0
        for _ in range(1 + i % 3):                          # pairs of one to three transcript segments
            n = int(rng.integers(2, 12))
            dur = float(rng.uniform(1.0, 6.0))
            segs.append({"start": t, "end": t + dur, "text": " ".join(rng.choice(words, n))})
            t += dur
        items.append((segs, np.zeros(int(t * 16000) + 800, dtype=np.float32)))

    def one(wav, toks):
        T, N = max(0, (len(wav) - 400) // 320 + 1), len(toks)
        if T < max(N, 2):
            return (T, None, None)
        idx = np.minimum((np.arange(T) * N) // T, N - 1)
        return (T, idx.tolist(), (0.25 + 0.5 * (np.arange(T) % 3 == 0)).tolist())

    def plain(wavs, toks, blank, beam):
        return [one(w, t) for w, t in zip(wavs, toks)]

    class Streaming:
        """hands the results out in forwards of `k` segments, longest first (any order must do)"""
        def __init__(self, k):
            self.k = k

        def __call__(self, wavs, toks, blank, beam):
            raise AssertionError("align_batch must take the streaming interface when there is one")

        def batches(self, wavs, toks, blank, beam):
            order = sorted(range(len(wavs)), key=lambda i: -len(wavs[i]))
            for a in range(0, len(order), self.k):
                idx = order[a: a + self.k]
                yield idx, [one(wavs[i], toks[i]) for i in idx]

    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        tr0 = []
        ref = AL.align_batch(items, None, meta, "cpu", _aligner=plain, _trace=tr0)
        for k in (1, 4, 100):
            tr = []
            got = AL.align_batch(items, None, meta, "cpu", _aligner=Streaming(k), _trace=tr)
            assert got == ref and tr == tr0, k
    assert sum(len(r["segments"]) for r in ref) >= 11 and any(s["words"] for r in ref for s in r["segments"])
    a = AL._HipAligner(None)
    for n in (1, 32, 33, 64, 65, 81, 129, 320):
        cuts = a._cuts(list(range(n)))
        assert [i for c in cuts for i in c] == list(range(n)) and max(len(c) for c in cuts) <= 64
        assert n <= 32 or (len(cuts) >= 2 and max(len(c) for c in cuts) - min(len(c) for c in cuts) <= 1)
