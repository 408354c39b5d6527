"""Forced alignment behind the reference's `align()` / `load_align_model()` API
(/root/reference/whisperx/alignment.py:77-110, :113-380).

Host side (this file): character cleaning, segment bookkeeping, char -> word -> sentence
assembly, NaN interpolation and the time rounding -- restated without pandas / nltk so
that the result dict is identical to the reference's.
Device side (libwxhip.so): the wav2vec2 CTC forward for a padded batch of segments
(wx_w2v_emissions, replaces the per-segment `model(waveform)` loop at :251-258 and its
"TODO batched inference" :240) and the trellis + beam backtrack (wx_ctc_align,
replaces get_trellis / backtrack_beam :268-269).
"""
import math
import os
import re
from typing import Iterable, List, Optional, Union

import numpy as np

from .audio import SAMPLE_RATE

PUNKT_ABBREVIATIONS = ['dr', 'vs', 'mr', 'mrs', 'prof']        # alignment.py:27
LANGUAGES_WITHOUT_SPACES = ["ja", "zh"]                          # alignment.py:29

DEFAULT_ALIGN_MODELS_TORCH = {                                   # alignment.py:31-37
    "en": "WAV2VEC2_ASR_BASE_960H",
    "fr": "VOXPOPULI_ASR_BASE_10K_FR",
    "de": "VOXPOPULI_ASR_BASE_10K_DE",
    "es": "VOXPOPULI_ASR_BASE_10K_ES",
    "it": "VOXPOPULI_ASR_BASE_10K_IT",
}
# HF equivalents of the torchaudio bundles (same architecture / training data); the HIP
# backend loads HF-format checkpoints from a local directory.
TORCHAUDIO_TO_HF = {"WAV2VEC2_ASR_BASE_960H": "facebook/wav2vec2-base-960h"}

DEFAULT_ALIGN_MODELS_HF = {                                      # alignment.py:39-74
    "ja": "jonatasgrosman/wav2vec2-large-xlsr-53-japanese",
    "zh": "jonatasgrosman/wav2vec2-large-xlsr-53-chinese-zh-cn",
    "nl": "jonatasgrosman/wav2vec2-large-xlsr-53-dutch",
    "uk": "Yehor/wav2vec2-xls-r-300m-uk-with-small-lm",
    "pt": "jonatasgrosman/wav2vec2-large-xlsr-53-portuguese",
    "ar": "jonatasgrosman/wav2vec2-large-xlsr-53-arabic",
    "cs": "comodoro/wav2vec2-xls-r-300m-cs-250",
    "ru": "jonatasgrosman/wav2vec2-large-xlsr-53-russian",
    "pl": "jonatasgrosman/wav2vec2-large-xlsr-53-polish",
    "hu": "jonatasgrosman/wav2vec2-large-xlsr-53-hungarian",
    "fi": "jonatasgrosman/wav2vec2-large-xlsr-53-finnish",
    "fa": "jonatasgrosman/wav2vec2-large-xlsr-53-persian",
    "el": "jonatasgrosman/wav2vec2-large-xlsr-53-greek",
    "tr": "mpoyraz/wav2vec2-xls-r-300m-cv7-turkish",
    "da": "saattrupdan/wav2vec2-xls-r-300m-ftspeech",
    "he": "imvladikon/wav2vec2-xls-r-300m-hebrew",
    "vi": 'nguyenvulebinh/wav2vec2-base-vi',
    "ko": "kresnik/wav2vec2-large-xlsr-korean",
    "ur": "kingabzpro/wav2vec2-large-xls-r-300m-Urdu",
    "te": "anuragshas/wav2vec2-large-xlsr-53-telugu",
    "hi": "theainerd/Wav2Vec2-large-xlsr-hindi",
    "ca": "softcatala/wav2vec2-large-xlsr-catala",
    "ml": "gvs/wav2vec2-large-xlsr-malayalam",
    "no": "NbAiLab/nb-wav2vec2-1b-bokmaal-v2",
    "nn": "NbAiLab/nb-wav2vec2-1b-nynorsk",
    "sk": "comodoro/wav2vec2-xls-r-300m-sk-cv8",
    "sl": "anton-l/wav2vec2-large-xlsr-53-slovenian",
    "hr": "classla/wav2vec2-xls-r-parlaspeech-hr",
    "ro": "gigant/romanian-wav2vec2",
    "eu": "stefan-it/wav2vec2-large-xlsr-53-basque",
    "gl": "ifrz/wav2vec2-large-xlsr-galician",
    "ka": "xsway/wav2vec2-large-xlsr-georgian",
    "lv": "jimregan/wav2vec2-large-xlsr-latvian-cv",
    "tl": "Khalsuu/filipino-wav2vec2-l-xls-r-300m-official",
}


def load_align_model(language_code: str, device: str, model_name: Optional[str] = None, model_dir=None):
    """alignment.py:77-110.  Returns (model, metadata) with metadata["type"] == "hip".
    Checkpoints are read from a local HF-format directory (no network on the GPU box):
    `model_dir/<model_name>` or `model_dir` itself."""
    import os
    from .w2v import W2VHipModel
    if model_name is None:
        if language_code in DEFAULT_ALIGN_MODELS_TORCH:
            model_name = DEFAULT_ALIGN_MODELS_TORCH[language_code]
        elif language_code in DEFAULT_ALIGN_MODELS_HF:
            model_name = DEFAULT_ALIGN_MODELS_HF[language_code]
        else:
            print(f"There is no default alignment model set for this language ({language_code}).\
                Please find a wav2vec2.0 model finetuned on this language in https://huggingface.co/models, then pass the model name in --align_model [MODEL_NAME]")
            raise ValueError(f"No default align-model for language: {language_code}")
    hf_name = TORCHAUDIO_TO_HF.get(model_name, model_name)
    candidates = [p for p in (
        model_name if os.path.isdir(str(model_name)) else None,
        os.path.join(model_dir, hf_name) if model_dir else None,
        os.path.join(model_dir, hf_name.split("/")[-1]) if model_dir else None,
        model_dir) if p and os.path.exists(os.path.join(p, "config.json"))]
    if not candidates:
        raise ValueError(f'The chosen align_model "{model_name}" could not be found locally (looked under '
                         f'model_dir={model_dir!r}); the HIP backend loads HF-format wav2vec2 checkpoints from disk')
    device_index = int(str(device).split(":")[1]) if ":" in str(device) else 0
    align_model, vocab = W2VHipModel.from_hf_dir(candidates[0], device_index=device_index)
    align_dictionary = {char.lower(): code for char, code in vocab.items()}
    align_metadata = {"language": language_code, "dictionary": align_dictionary, "type": "hip"}
    return align_model, align_metadata


# --------------------------------------------------------------------------- sentence spans
_PERIOD_CONTEXT = re.compile(r"\S*[.?!](?=(?P<after_tok>[?!)\";}\]\*:@'\({\[])|\s+(?P<next_tok>\S+))")
_CLOSERS = ")\"'”’]}"


def sentence_spans(text: str, abbreviations=PUNKT_ABBREVIATIONS):
    """Approximation of the untrained PunktSentenceTokenizer(abbrev_types=...).span_tokenize
    the reference builds at alignment.py:191-194 (nltk is not a dependency here): a token
    ending in . ? ! followed by whitespace ends a sentence unless it is a listed
    abbreviation, an ellipsis, a single-letter initial before a word, or a number before a
    lower-case word.  Spans exclude the separating whitespace."""
    abbrevs = set(abbreviations)
    spans, start = [], 0
    for m in _PERIOD_CONTEXT.finditer(text):
        tok = m.group(0)
        end = m.end()
        if tok.endswith("."):
            word = tok[:-1].lstrip("\"'([{“‘").lower()
            nxt = m.group("next_tok") or ""
            if word.endswith(".."):
                continue
            if word in abbrevs or word.split("-")[-1] in abbrevs:
                continue
            if len(word) == 1 and word.isalpha() and nxt[:1].isalpha():
                continue
            if re.fullmatch(r"[\d.,]+", word or "x") and nxt[:1].islower():
                continue
        while end < len(text) and text[end] in _CLOSERS:
            end += 1
        if end >= len(text) or not text[end].isspace():
            if m.group("after_tok") is None:
                continue
        if end > start and text[start:end].strip():
            spans.append((start, end))
        j = end
        while j < len(text) and text[j].isspace():
            j += 1
        start = j
    if start < len(text) and text[start:].strip():
        spans.append((start, len(text.rstrip()) if text.rstrip() else len(text)))
    if not spans and text:
        spans = [(0, len(text))]
    return spans


# --------------------------------------------------------------------------- small pandas stand-ins
def _nanmin(vals):
    v = [x for x in vals if x is not None and not (isinstance(x, float) and math.isnan(x))]
    return min(v) if v else float("nan")


def _nanmax(vals):
    v = [x for x in vals if x is not None and not (isinstance(x, float) and math.isnan(x))]
    return max(v) if v else float("nan")


def _nanmean(vals):
    v = [x for x in vals if x is not None and not (isinstance(x, float) and math.isnan(x))]
    return float(np.mean(np.asarray(v, dtype=np.float64))) if v else float("nan")


def _round3(x: np.ndarray) -> list:
    """[round(v, 3) for v in x] -- Python's round (the nearest k / 1000 to the exact value of the double, as the reference
    computes it, alignment.py:288-290), vectorised: k = rint(x * 1000) is that nearest k unless the product's own rounding
    moved it across a half: y = fl(x * 1000) lies within half an ulp of the exact product, so that needs a half-integer
    within half an ulp of y; elements within TWO ulps of one (about one in 10^12, whatever the magnitude: absolute times
    of a whole file included) go through Python's round.  k / 1000.0 is the correctly rounded quotient, the double round() returns."""
    x = np.asarray(x, dtype=np.float64)
    y = x * 1000.0
    k = np.rint(y)
    out = (k / 1000.0).tolist()
    with np.errstate(invalid="ignore"):
        near = np.flatnonzero(~(np.abs(np.abs(y - k) - 0.5) > 2.0 * np.spacing(np.abs(y))) | ~np.isfinite(y))
    for i in near.tolist():
        out[i] = round(float(x[i]), 3)
    return out


def _mean_f64(v):
    """np.mean of a list of Python floats, bit for bit, without its per-call overhead for the short lists a word's
    characters make: below 8 elements numpy's pairwise sum is the plain left-to-right sum"""
    n = len(v)
    if n < 8:
        t = 0.0
        for x in v:
            t += x
        return t / n
    return float(np.add.reduce(np.asarray(v, dtype=np.float64)) / n)


def interpolate_nans(x: List[float], method="nearest"):
    """whisperx/utils.py:438-442 for a float column: nearest-valid fill inside the valid
    range (scipy 'nearest': ties go to the lower index), then ffill / bfill."""
    x = [float("nan") if v is None else float(v) for v in x]
    valid = [i for i, v in enumerate(x) if not math.isnan(v)]
    out = list(x)
    if len(valid) > 1:
        if method == "nearest":
            for i, v in enumerate(x):
                if math.isnan(v) and valid[0] < i < valid[-1]:
                    lo = max(j for j in valid if j < i)
                    hi = min(j for j in valid if j > i)
                    out[i] = x[lo] if (i - lo) <= (hi - i) else x[hi]
        elif method == "linear":
            for i, v in enumerate(x):
                if math.isnan(v) and valid[0] < i < valid[-1]:
                    lo = max(j for j in valid if j < i)
                    hi = min(j for j in valid if j > i)
                    out[i] = x[lo] + (x[hi] - x[lo]) * (i - lo) / (hi - lo)
        elif method not in ("ignore",):
            raise ValueError(f"unsupported interpolate_method {method!r}")
    last = float("nan")
    for i in range(len(out)):            # ffill
        if math.isnan(out[i]):
            out[i] = last
        else:
            last = out[i]
    nxt = float("nan")
    for i in range(len(out) - 1, -1, -1):  # bfill
        if math.isnan(out[i]):
            out[i] = nxt
        else:
            nxt = out[i]
    return out


def merge_repeats(path_tok, path_score, transcript):
    """alignment.py:597-613 on the device path arrays -> [(label, start, end, score)]: runs of equal token index, the
    score of a run = the plain left-to-right sum of its frame scores over its length (run boundaries found vectorised)."""
    n = len(path_tok)
    if n == 0:
        return []
    tok = np.asarray(path_tok)
    cuts = [0] + (np.flatnonzero(tok[1:] != tok[:-1]) + 1).tolist() + [n]
    if not isinstance(path_score, list):
        path_score = [float(x) for x in path_score]
    segs = []
    for i1, i2 in zip(cuts[:-1], cuts[1:]):
        t = 0.0
        for x in path_score[i1:i2]:
            t += x
        segs.append((transcript[path_tok[i1]], i1, i2, t / (i2 - i1)))
    return segs


class _HipAligner:
    """Default numeric backend of align(): batched wav2vec2 emissions + CTC DP on the GPU."""

    def __init__(self, model, max_batch=64):
        # segments per wav2vec2 forward (sorted by length, padded to the longest of the batch): the 768-wide GEMMs are
        # 18 tiles of 256 x 256 per 30 s segment, 288 for 16 segments on 256 CUs = two rounds of which the second is
        # nearly empty; 64 segments make 4.5 rounds (forward 13.9 -> 11.7 ms per 16 x 30 s, tools/probe_w2v.py)
        self.model = model
        self.max_batch = max_batch

    def __call__(self, waveforms, token_lists, blank_id, beam_width=2):
        """waveforms: list of 1-D float32 numpy; returns per segment (T, path_tok, path_score) or None."""
        results = [None] * len(waveforms)
        for idx, res in self.batches(waveforms, token_lists, blank_id, beam_width):
            for i, r in zip(idx, res):
                results[i] = r
        return results

    def _cuts(self, order):
        """the forwards a job is cut into: up to max_batch segments each; a job of more than 32 segments goes as at least two
        forwards of about equal size, so that the host can assemble one forward's words while the GPU runs the next"""
        n = len(order)
        if n <= 32:
            return [order] if n else []
        k = max(2, -(-n // self.max_batch), -(-n // 32) if n <= 2 * self.max_batch else 0)
        k = min(k, -(-n // 16))
        return [order[(n * j) // k: (n * (j + 1)) // k] for j in range(k)]

    def _submit(self, idx, waveforms, token_lists, blank_id, beam_width, slot):
        """one forward + CTC DP enqueued on the model's stream, inputs and results through pinned host buffers: nothing here
        waits for the GPU.  Everything is issued ON the model's stream (no cross-stream waits, no pageable copies: a
        `tensor.to(device)` from pageable memory blocks the host until the stream has drained, i.e. until the PREVIOUS
        forward has finished -- 7 ms a call, which is what the first version of this pipeline spent its overlap on)."""
        import torch
        m = self.model
        batch = [waveforms[i] for i in idx]
        S = len(idx)
        Nmax = max(len(token_lists[i]) for i in idx)
        # two sets of pinned buffers (one forward being read while the next is in flight), grown on demand and kept with the
        # MODEL: an aligner object lives for one align() call, and pinning host memory costs milliseconds per buffer
        bufs = m.__dict__.setdefault("_aligner_pinned", [None, None])
        with torch.cuda.stream(m.stream):
            if all(torch.is_tensor(w) and w.is_cuda for w in batch):
                # audio already resident in HBM (transcribe_batch on device tensors): the padded batch is built on the
                # device, nothing crosses PCIe
                n = [max(int(w.shape[0]), 400) for w in batch]
                pcm = torch.zeros(S, max(n), dtype=torch.float32, device=batch[0].device)
                for r, w in enumerate(batch):
                    pcm[r, : w.shape[0]] = w
                logp, T = m.emissions_device(pcm, n)
            else:
                logp, T = m.emissions([w.detach().cpu().numpy() if torch.is_tensor(w) else w for w in batch])
            Tmax = logp.shape[1]
            # FLAT pinned buffers, viewed at exactly this forward's shape: a copy between a device tensor and a NON-contiguous
            # pinned view (a [:S, :Tmax] corner of a larger buffer) is staged and synchronous in torch -- the host then sits out
            # the whole forward inside "non_blocking" copies (19-23 ms per forward of 27 segments, the pipeline serialised)
            cur = bufs[slot]
            if cur is None or cur[0].numel() < S * Tmax or cur[3].numel() < S * Nmax or cur[2].numel() < S:
                n_out, n_in, rows = max(S * Tmax, self.max_batch * 1500), max(S * Nmax, self.max_batch * 2048), max(S, self.max_batch)
                pin = lambda n, dtype=torch.int32: torch.empty(n, dtype=dtype).pin_memory()   # noqa: E731
                cur = bufs[slot] = (pin(n_out), pin(n_out, torch.float32), pin(rows), pin(n_in), pin(rows), pin(rows))
            h_tok, h_score, h_ok = cur[0][: S * Tmax].view(S, Tmax), cur[1][: S * Tmax].view(S, Tmax), cur[2][:S]
            i_tok, i_N, i_T = cur[3][: S * Nmax].view(S, Nmax), cur[4][:S], cur[5][:S]
            i_tok.zero_()
            for r, i in enumerate(idx):
                n_i = len(token_lists[i])
                i_tok[r, :n_i] = torch.tensor(token_lists[i], dtype=torch.int32)
                i_N[r] = n_i
                i_T[r] = T[r]
            dev = logp.device
            d_tok = i_tok.to(dev, non_blocking=True)
            d_N = i_N.to(dev, non_blocking=True)
            d_T = i_T.to(dev, non_blocking=True)
            ptok, pscore, ok, _ = m.ctc_align(logp, d_T, d_tok, d_N, blank_id, beam_width)
            h_tok.copy_(ptok, non_blocking=True)
            h_score.copy_(pscore, non_blocking=True)
            h_ok.copy_(ok, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(m.stream)
        return idx, T, h_tok, h_score, h_ok, ev, (ptok, pscore, ok, logp, d_tok, d_N, d_T)      # (the device tensors stay alive until collected)

    @staticmethod
    def _collect(handle):
        idx, T, h_tok, h_score, h_ok, ev, _keep = handle
        ev.synchronize()
        ptok, pscore, ok = h_tok.numpy(), h_score.numpy(), h_ok.numpy()
        return idx, [(T[r], ptok[r, : T[r]].tolist(), pscore[r, : T[r]].tolist()) if ok[r] else (T[r], None, None)
                     for r in range(len(idx))]

    def batches(self, waveforms, token_lists, blank_id, beam_width=2):
        """yields (segment indices, their results) forward by forward, shortest segments first.  The next forward is
        already enqueued when one is handed out: the caller's host work on a forward's words (align_batch: char -> word ->
        sentence assembly, the larger half of config 4's alignment stage) runs beside the GPU's work on the next."""
        order = sorted(range(len(waveforms)), key=lambda i: len(waveforms[i]))
        pending = None
        for k, idx in enumerate(self._cuts(order)):
            handle = self._submit(idx, waveforms, token_lists, blank_id, beam_width, k & 1)
            if pending is not None:
                yield self._collect(pending)
            pending = handle
        if pending is not None:
            yield self._collect(pending)


def word_index(text: str, model_lang: str = "en") -> List[int]:
    """the word number of every character of a segment text, as align() counts them (alignment.py:296-309): the number
    grows behind a character that is followed by a space (behind every character for languages without spaces)"""
    no_spaces = model_lang in LANGUAGES_WITHOUT_SPACES
    n = len(text)
    idx = [0] * n
    w = 0
    for cdx in range(n):
        idx[cdx] = w
        if no_spaces or cdx == n - 1 or text[cdx + 1] == " ":
            w += 1
    return idx


def sentence_word_texts(text: str, sstart: int, send: int, model_lang: str = "en", widx: Optional[List[int]] = None):
    """the words of one sentence span of a segment text, as align() forms them (alignment.py:296-343): the sentence takes the
    character rows sstart..send INCLUSIVE (pandas .loc), a word is the run of its characters that share a word number
    (word_index), its text those characters joined and stripped, empty ones are dropped.  Returns [(word number, word
    text)]: what a rank that holds only a record needs to rebuild the words.  `widx`: word_index(text) when the caller
    has it (one pass over the text for all of its sentences)."""
    if widx is None:
        widx = word_index(text, model_lang)
    lo, hi = sstart, min(send + 1, len(text))
    out = []
    i = lo
    while i < hi:
        j = i + 1
        while j < hi and widx[j] == widx[i]:
            j += 1
        wt = text[i:j].strip()
        if wt:
            out.append((widx[i], wt))
        i = j
    return out


def align(
    transcript: Iterable[dict],
    model,
    align_model_metadata: dict,
    audio: Union[str, np.ndarray],
    device: str,
    interpolate_method: str = "nearest",
    return_char_alignments: bool = False,
    print_progress: bool = False,
    combined_progress: bool = False,
    _aligner=None,
    _sentence_spans=None,
) -> dict:
    """Align phoneme recognition predictions to known transcription (alignment.py:113-380).

    `_aligner` / `_sentence_spans` are injection points for the CPU tests (golden emissions,
    the fixture's sentence spans); the defaults are the HIP backend and `sentence_spans`."""
    return align_batch([(transcript, audio)], model, align_model_metadata, device, interpolate_method=interpolate_method,
                       return_char_alignments=return_char_alignments, print_progress=print_progress,
                       combined_progress=combined_progress, _aligner=_aligner, _sentence_spans=_sentence_spans)[0]


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
        _aligner = _HipAligner(model)
    span_fn = _sentence_spans or (lambda sdx, text: sentence_spans(text))
    blank_id = 0
    for char, code in model_dictionary.items():
        if char == '[pad]' or char == '<pad>':
            blank_id = code

    prepared = []
    char_map = {}
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
            # per character: lower-case, " " -> "|" (languages with spaces), "*" for anything the model's dictionary does
            # not hold; leading / trailing whitespace is skipped (alignment.py:157-176).  One lookup per DISTINCT character.
            a, b = num_leading, len(text) - num_trailing
            clean_char = []
            for char in text[a:b]:
                c = char_map.get(char)
                if c is None:
                    c = char.lower()
                    if model_lang not in LANGUAGES_WITHOUT_SPACES:
                        c = c.replace(" ", "|")
                    if c not in model_dictionary:
                        c = '*'
                    char_map[char] = c
                clean_char.append(c)
            clean_cdx = list(range(a, max(a, b)))
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

    # the aligner hands its results out forward by forward when it can (_HipAligner.batches: the next forward already runs
    # on the GPU); a pair is assembled as soon as the last of its segments is back
    by_key = {}
    out: List[Optional[dict]] = [None] * len(prepared)
    traces: List[Optional[list]] = [None] * len(prepared)
    waiting = [0] * len(prepared)
    for j in jobs:
        waiting[j[0]] += 1

    def assemble(pi):
        transcript, segment_data, MAX_DURATION = prepared[pi]
        traces[pi] = [] if _trace is not None else None
        out[pi] = _assemble(pi, transcript, segment_data, MAX_DURATION, by_key, model_lang, interpolate_method,
                            return_char_alignments, traces[pi])

    wavs, toks = [j[2] for j in jobs], [j[3] for j in jobs]
    # the assembly allocates a few hundred thousand small objects and no reference cycles: with the collector left on, every
    # other stage ran into a full collection of the process's heap (the tokenizer caches, the previous results) and took
    # 100 ms instead of 50 (tools/prof_align_stage.py)
    import gc
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        _run_aligner_and_assemble(_aligner, jobs, wavs, toks, blank_id, by_key, waiting, assemble)
    finally:
        if gc_was_on:
            gc.enable()
    for pi in range(len(prepared)):
        if out[pi] is None:                    # pairs none of whose segments went to the aligner
            assemble(pi)
    if _trace is not None:
        _trace.extend(traces)
    return out


def _run_aligner_and_assemble(_aligner, jobs, wavs, toks, blank_id, by_key, waiting, assemble):
    if jobs:
        stream = _aligner.batches(wavs, toks, blank_id, 2) if hasattr(_aligner, "batches") else \
            [(list(range(len(jobs))), _aligner(wavs, toks, blank_id, 2))]
        for idx, res in stream:
            for i, r in zip(idx, res):
                j = jobs[i]
                by_key[(j[0], j[1])] = (r, j[4])
                waiting[j[0]] -= 1
                if waiting[j[0]] == 0:
                    assemble(j[0])


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

        # assign timestamps to aligned characters (alignment.py:281-309).  The reference keeps a pandas frame of one row per
        # character and filters it per sentence and per word (:296-343); here the columns are plain lists and a word is the
        # contiguous run of characters that share a word index (the index only grows along the text), so a segment costs
        # O(characters) instead of O(words x characters) -- this loop is the host half of config 4's alignment stage
        clean_cdx = segment_data[sdx]["clean_cdx"]
        n_text = len(text)
        st: List[Optional[float]] = [None] * n_text
        en: List[Optional[float]] = [None] * n_text
        sc: List[Optional[float]] = [None] * n_text
        # round(x, 3) of every character's start / end / score in three vectorised calls (_round3: Python's rounding, exactly)
        n_cs = len(clean_cdx)
        cs_arr = np.asarray([(c[1], c[2]) for c in char_segments[:n_cs]], dtype=np.float64).reshape(-1, 2)
        st_v = _round3(cs_arr[:, 0] * ratio + t1)
        en_v = _round3(cs_arr[:, 1] * ratio + t1)
        sc_v = _round3(np.asarray([c[3] for c in char_segments[:n_cs]], dtype=np.float64))
        for pos, cdx in enumerate(clean_cdx):
            st[cdx], en[cdx], sc[cdx] = st_v[pos], en_v[pos], sc_v[pos]
        no_spaces = model_lang in LANGUAGES_WITHOUT_SPACES
        widx = [0] * n_text
        w = 0
        for cdx in range(n_text):
            widx[cdx] = w
            if no_spaces or cdx == n_text - 1 or text[cdx + 1] == " ":
                w += 1

        aligned_subsegments = []
        scored: List[dict] = []                  # the segment's words that carry a score (rounded together afterwards)
        for sstart, send in segment_data[sdx]["sentence_spans"]:
            lo, hi = sstart, min(send + 1, n_text)          # pandas .loc is end-inclusive (alignment.py:317)
            sentence_text = text[sstart:send]
            v = [x for x in st[lo:hi] if x is not None]
            sentence_start = min(v) if v else float("nan")
            v = [en[c] for c in range(lo, hi) if en[c] is not None and text[c] != ' ']
            sentence_end = max(v) if v else float("nan")
            sentence_words = []
            i = lo
            while i < hi:
                j = i + 1
                while j < hi and widx[j] == widx[i]:
                    j += 1
                word_text = text[i:j].strip()
                if len(word_text) != 0:
                    cs = [c for c in range(i, j) if text[c] != " "]
                    v = [st[c] for c in cs if st[c] is not None]
                    word_start = min(v) if v else float("nan")
                    v = [en[c] for c in cs if en[c] is not None]
                    word_end = max(v) if v else float("nan")
                    v = [sc[c] for c in cs if sc[c] is not None]
                    word_segment = {"word": word_text}
                    if not math.isnan(word_start):
                        word_segment["start"] = word_start
                    if not math.isnan(word_end):
                        word_segment["end"] = word_end
                    if v:
                        # pandas .mean() yields np.float64 (numpy's pairwise sum), whose round() is numpy's: scale, rint, unscale --
                        # applied to all of the segment's words at once below (np.round of an array is that round() per element)
                        word_segment["score"] = _mean_f64(v)
                        scored.append(word_segment)
                    sentence_words.append(word_segment)
                i = j
            sub = {"text": sentence_text, "start": sentence_start, "end": sentence_end, "words": sentence_words,
                   "_span": (sstart, send)}
            if return_char_alignments:
                chars = []
                for c_ in range(lo, hi):
                    c = {"char": text[c_]}
                    for key, col in (("start", st), ("end", en), ("score", sc)):
                        if col[c_] is not None and col[c_] != -1:
                            c[key] = col[c_]
                    chars.append(c)
                sub["chars"] = chars
            aligned_subsegments.append(sub)

        if scored:
            for w_, r_ in zip(scored, np.round(np.asarray([w_["score"] for w_ in scored], dtype=np.float64), 3).tolist()):
                w_["score"] = r_
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
