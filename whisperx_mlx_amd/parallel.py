"""Multi-GPU sharding of the chunk list.  Every <= 30 s chunk is independent end
to end (the reference assumes the same: whisperx/asr.py:70-87, condition_on_previous_text
False at whisperx/backends/mlx_whisper.py:79), so chunks are dealt to ranks with no
data-path collective and the fixed-width result records come back with ONE all_gather
(RCCL over xGMI when the process group is "nccl"; gloo on CPU for the tests).  Every rank
knows the whole chunk list, hence every rank's share: no size exchange precedes the gather."""
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

MAX_TOK = 224          # sample_len
# int32 record: [chunk_id, n_tokens, sum_logprob bits, no_speech bits, tokens[224], n_words,
#                            word_tok_end[224], word_start_ms[224], word_end_ms[224],
#   -- the wav2vec2-aligned words of the chunk (config 4: the alignment stage shards like the chunks do and its result
#      travels in the SAME gather; /root/reference/whisperx/alignment.py:206-373 aligns every segment independently) --
#                            n_out (-1: no alignment payload), n_aligned_words, n_sentences,
#                            out[48] x (kind 0 failed / 1 aligned, transcript segment, start_ms, end_ms),
#                            sentence[96] x (out index, span begin, span end)   spans index the segment's text,
#                            aligned_word[224] x (start_ms, end_ms, score_milli)   MISSING where align() omits the key]
_O_WORDS = 4 + MAX_TOK
_O_ALIGN = _O_WORDS + 1 + 3 * MAX_TOK
MAX_OUT, MAX_SENT = 48, 96
_O_OUT = _O_ALIGN + 3
_O_SENT = _O_OUT + 4 * MAX_OUT
_O_AWORD = _O_SENT + 3 * MAX_SENT
REC_W = _O_AWORD + 3 * MAX_TOK
MISSING = -2 ** 31


def shard_indices(durations: Sequence[float], rank: int, world: int) -> List[int]:
    """Longest-first round-robin deal: balances decode length across ranks."""
    order = sorted(range(len(durations)), key=lambda i: (-durations[i], i))
    return sorted(order[rank::world])


def word_spans(result: Dict) -> List[tuple]:
    """(tok_end, start ms, end ms) per word of a per-chunk result of WhisperHipBackend._decode_chunks.  tok_end = index one
    past the word's last token in the chunk's TEXT ids -- the record's `tokens` with the timestamp tokens (ids >=
    timestamp_begin) taken out -- so a rank that holds only the record rebuilds word text and times.  The backend emits
    `word_tok_end` from the loop that builds `words` (whitespace-only words are dropped there and their tokens count
    towards the next word); `word_token_counts` (one count per kept word) is the older form."""
    words = result.get("words", [])
    ends = result.get("word_tok_end")
    if ends is None:
        ends, pos = [], 0
        for n in result.get("word_token_counts") or [1] * len(words):
            pos += int(n)
            ends.append(pos)
    assert len(ends) == len(words), "one token position per word"
    return [(int(e), int(round(w["start"] * 1000)), int(round(w["end"] * 1000))) for w, e in zip(words, ends)]


def pack_records(results: List[Dict], chunk_ids: Sequence[int]) -> torch.Tensor:
    rec = np.zeros((len(results), REC_W), dtype=np.int32)
    fbits = rec[:, 2:4].view(np.float32)          # sum_logprob, no_speech_prob as float32 bit patterns
    for r, (res, cid) in enumerate(zip(results, chunk_ids)):
        toks = res["tokens"][:MAX_TOK]
        row = rec[r]
        row[0], row[1] = int(cid), len(toks)
        fbits[r, 0], fbits[r, 1] = res.get("sum_logprob", res.get("avg_logprob", 0.0)), res.get("no_speech_prob", 0.0)
        row[4: 4 + len(toks)] = toks
        cols = res.get("word_spans_np")              # (tok_end, start_ms, end_ms) int32 arrays straight from the backend's host half
        if cols is not None and res.get("word_spans") is None:
            nw = min(len(cols[0]), MAX_TOK)
            row[_O_WORDS] = nw
            for k in range(3):
                row[_O_WORDS + 1 + k * MAX_TOK: _O_WORDS + 1 + k * MAX_TOK + nw] = cols[k][:nw]
        else:
            words = res.get("word_spans")
            if words is None:
                words = word_spans(res)
            words = words[:MAX_TOK]
            row[_O_WORDS] = len(words)
            if words:
                w = np.asarray(words, dtype=np.int32)         # (n_words, 3): tok_end, start_ms, end_ms
                for k in range(3):
                    row[_O_WORDS + 1 + k * MAX_TOK: _O_WORDS + 1 + k * MAX_TOK + len(words)] = w[:, k]
        row[_O_ALIGN] = -1
        if res.get("aligned") is not None:
            pack_aligned(row, *res["aligned"])
    return torch.from_numpy(rec)


def _ms(x: float) -> int:
    """a time align() rounded to 3 decimals (alignment.py:288-290) as integer milliseconds -- exact both ways: k / 1000.0
    is the double round(x, 3) returned.  Anything else (interpolate_method="linear" sentence times) does not travel."""
    k = int(round(x * 1000.0))
    if k / 1000.0 != x:
        raise ValueError(f"aligned time {x!r} is not a whole number of milliseconds: the multi-GPU record carries align()'s "
                         f"3-decimal times only (interpolate_method='nearest')")
    return k


def pack_aligned(row: np.ndarray, aligned: Dict, trace: List[tuple]) -> None:
    """one align() result (relative to its VAD segment) + the trace alignment.align_batch kept of it -> the record's
    alignment block.  Word texts and segment texts do not travel: the receiver rebuilds them from the chunk's own text
    and the sentence spans (alignment.sentence_word_texts)."""
    segs = aligned["segments"]
    assert len(segs) == len(trace), (len(segs), len(trace))
    if len(segs) > MAX_OUT:
        raise ValueError(f"{len(segs)} aligned segments in one chunk (record holds {MAX_OUT})")
    n_sent = n_w = 0
    for o, (seg, tr) in enumerate(zip(segs, trace)):
        base = _O_OUT + 4 * o
        if tr[0] == "fail":
            row[base: base + 4] = (0, tr[1], 0, 0)              # start / end are the transcript segment's own (the receiver has them)
            continue
        row[base: base + 4] = (1, tr[1], _ms(seg["start"]), _ms(seg["end"]))
        for (sb, se) in tr[2]:
            if n_sent >= MAX_SENT:
                raise ValueError(f"more than {MAX_SENT} sentences in one chunk")
            row[_O_SENT + 3 * n_sent: _O_SENT + 3 * n_sent + 3] = (o, sb, se)
            n_sent += 1
        for w in seg["words"]:
            if n_w >= MAX_TOK:
                raise ValueError(f"more than {MAX_TOK} aligned words in one chunk")
            row[_O_AWORD + 3 * n_w: _O_AWORD + 3 * n_w + 3] = (_ms(w["start"]) if "start" in w else MISSING,
                                                               _ms(w["end"]) if "end" in w else MISSING,
                                                               _ms(w["score"]) if "score" in w else MISSING)
            n_w += 1
    row[_O_ALIGN: _O_ALIGN + 3] = (len(segs), n_w, n_sent)


def unpack_aligned(row: np.ndarray) -> Optional[Dict]:
    n_out = int(row[_O_ALIGN])
    if n_out < 0:
        return None
    out = [tuple(int(v) for v in row[_O_OUT + 4 * o: _O_OUT + 4 * o + 4]) + ([],) for o in range(n_out)]
    for k in range(int(row[_O_ALIGN + 2])):
        o, sb, se = (int(v) for v in row[_O_SENT + 3 * k: _O_SENT + 3 * k + 3])
        out[o][4].append((sb, se))
    n_w = int(row[_O_ALIGN + 1])
    words = [tuple(int(v) for v in row[_O_AWORD + 3 * k: _O_AWORD + 3 * k + 3]) for k in range(n_w)]
    return {"out": out, "words": words}


def assemble_aligned(rel_segments: List[Dict], payload: Dict, model_lang: str = "en") -> Dict:
    """the align() result dict of one VAD segment from its record: rel_segments = the transcript segments that were
    aligned (relative times and text, rebuilt by the receiver from the chunk records), payload = unpack_aligned()."""
    from .alignment import LANGUAGES_WITHOUT_SPACES, sentence_word_texts, word_index
    joiner = "".join if model_lang in LANGUAGES_WITHOUT_SPACES else " ".join
    segs, wi = [], 0
    widx_of = {}                                  # word numbers of a segment text, computed once for all of its sentences
    for kind, sdx, start_ms, end_ms, spans in payload["out"]:
        src = rel_segments[sdx]
        if kind == 0:
            segs.append({"start": src["start"], "end": src["end"], "text": src["text"], "words": [], "chars": None})
            continue
        text = src["text"]
        if sdx not in widx_of:
            widx_of[sdx] = word_index(text, model_lang)
        words = []
        for sb, se in spans:
            for _widx, wt in sentence_word_texts(text, sb, se, model_lang, widx_of[sdx]):
                s_ms, e_ms, sc = payload["words"][wi]
                wi += 1
                w = {"word": wt}
                if s_ms != MISSING:
                    w["start"] = s_ms / 1000.0
                if e_ms != MISSING:
                    w["end"] = e_ms / 1000.0
                if sc != MISSING:
                    w["score"] = sc / 1000.0
                words.append(w)
        segs.append({"start": start_ms / 1000.0, "end": end_ms / 1000.0, "text": joiner(text[sb:se] for sb, se in spans),
                     "words": words})
    assert wi == len(payload["words"]), (wi, len(payload["words"]))
    word_segments = [w for s_ in segs for w in s_["words"]]
    return {"segments": segs, "word_segments": word_segments}


def unpack_records(rec: torch.Tensor) -> List[Dict]:
    out = []
    rec = rec.cpu().numpy()
    for row in rec:
        n = int(row[1])
        nw = int(row[_O_WORDS])
        f = row[2:4].view(np.float32)
        cols = [row[_O_WORDS + 1 + k * MAX_TOK: _O_WORDS + 1 + k * MAX_TOK + nw].tolist() for k in range(3)]
        out.append({"chunk_id": int(row[0]), "tokens": row[4: 4 + n].tolist(), "sum_logprob": float(f[0]),
                    "no_speech_prob": float(f[1]), "word_spans": list(zip(*cols)) if nw else [],
                    "aligned": unpack_aligned(row)})
    return out


def gather_records(local: torch.Tensor, counts: Optional[Sequence[int]] = None, device=None) -> List[Dict]:
    """THE collective: one all_gather of the fixed-width records, padded to the largest share; every rank returns
    the full list ordered by chunk_id.  `counts[r]` = records rank r contributes (computable on every rank from
    shard_indices, so no size exchange).  counts=None (a caller that does not know the other ranks' shares): the
    shares are exchanged first with one 8-byte all_gather -- a second collective, so the product paths always pass
    `counts`."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return sorted(unpack_records(local), key=lambda r: r["chunk_id"])
    world = dist.get_world_size()
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device())
                                             if dist.get_backend() == "nccl" else torch.device("cpu"))
    if counts is None:
        sizes = torch.zeros(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(sizes, torch.tensor([local.shape[0]], dtype=torch.int64, device=dev))
        counts = [int(v) for v in sizes.tolist()]
    n_max = int(max(counts))
    assert local.shape[0] == counts[dist.get_rank()], "gather_records: counts[rank] must be this rank's number of records"
    pad = torch.full((n_max, REC_W), -1, dtype=torch.int32, device=dev)
    pad[: local.shape[0]] = local.to(dev)
    allr = torch.empty(world * n_max, REC_W, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(allr, pad)
    allr = allr[allr[:, 0] >= 0]
    return sorted(unpack_records(allr), key=lambda r: r["chunk_id"])


def transcribe_sharded(backend, chunks: List[np.ndarray], language=None, task="transcribe", word_timestamps=False):
    """Each rank decodes its shard of `chunks` on its own GPU, then one gather.  With language=None the language is
    detected ONCE, from the first chunk of the whole list (every rank holds it and runs the same deterministic
    kernels), so all ranks decode with the same prompt."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if language is None:
        language = backend.detect_language(chunks[0]) if (chunks and backend.is_multilingual) else "en"
    lens = [len(c) for c in chunks]
    shares = [shard_indices(lens, r, world) for r in range(world)]
    mine = shares[rank]
    results = backend._decode_chunks([chunks[i] for i in mine], language, task, word_timestamps) if mine else []
    for r in results:
        r.setdefault("sum_logprob", r["avg_logprob"] * (len(r["tokens"]) + 1))
    return gather_records(pack_records(results, mine), counts=[len(s) for s in shares])


def transcribe_batch_sharded(backend, segments: List[Dict], batch_size: int = 16, align_words: bool = False, language=None,
                             task: str = "transcribe", word_timestamps=False, **kw) -> Dict:
    """`WhisperHipBackend.transcribe_batch` over the GPUs of one node (BASELINE.json config 4 as written: large-v3 +
    wav2vec2 align_model, VAD chunks sharded over the ranks, one RCCL gather): every rank transcribes -- and, with
    align_words, force-aligns (its own W2VHipModel) -- its share of the VAD segments, the fixed-width records (token ids,
    log-probabilities, DTW word spans, aligned words) come back with ONE all_gather, and every rank returns the dict
    the single-process call returns.

    segments: [{"start", "end", "audio"}] as whisperx/asr.py:70-73 builds them from the VAD's merged chunks, each <= 30 s
    (Vad.merge_chunks guarantees it for chunk_size 30; one segment = one chunk = one record).  The reference aligns
    every segment independently (/root/reference/whisperx/alignment.py:206-373, driven per VAD segment by
    /root/reference/whisperx/backends/mlx_lightning.py:290-369), so the alignment stage shards exactly like the ASR."""
    import torch.distributed as dist
    from .audio import N_SAMPLES, SAMPLE_RATE
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    has_audio = [s.get("audio") is not None for s in segments]           # (segments without audio are skipped, as transcribe_batch does)
    lens = [int(len(s["audio"])) if ok else 0 for s, ok in zip(segments, has_audio)]
    if any(n > N_SAMPLES for n in lens):
        raise ValueError("transcribe_batch_sharded: segments must be <= 30 s (one chunk each); split longer ones first")
    if language is None:
        first = next((s["audio"] for s, ok in zip(segments, has_audio) if ok), None)
        language = backend.detect_language(first) if (first is not None and backend.is_multilingual) else "en"
    shares = [shard_indices(lens, r, world) for r in range(world)]
    mine = [i for i in shares[rank] if has_audio[i]]
    counts = [sum(1 for i in sh if has_audio[i]) for sh in shares]
    my_segments = [segments[i] for i in mine]
    if kw.get("forced_lens") is not None:            # bench workload: one forced length per segment of the WHOLE list
        kw = dict(kw, forced_lens=[kw["forced_lens"][i] for i in mine])
    dtw = word_timestamps if word_timestamps in ("dtw", "dtw_inrepo") else (word_timestamps is True and not align_words)
    chunks = []
    if my_segments:
        res = backend.transcribe_batch(my_segments, batch_size=batch_size, language=language, task=task, word_timestamps=dtw,
                                       return_chunks=True, align_words=False, **kw)
        chunks = res["chunks"]
        assert [c["segment"] for c in chunks] == list(range(len(my_segments)))
        if align_words:
            groups = backend._group_by_vad(res["segments"], my_segments)
            trace: List[list] = []
            aligned = backend.align_groups(groups, my_segments, language, _trace=trace)
            for (vi, _rel), a, tr in zip(groups, aligned, trace):
                chunks[vi]["aligned"] = (a, tr)
    for c in chunks:
        c.setdefault("sum_logprob", c["avg_logprob"] * (len(c["tokens"]) + 1))
    records = gather_records(pack_records(chunks, mine), counts=counts)
    return assemble_result(backend.tokenizer, segments, records, language, dtw=bool(dtw), align_words=align_words)


def assemble_result(tokenizer, segments: List[Dict], records: List[Dict], language: str, dtw: bool = False,
                    align_words: bool = False) -> Dict:
    """the dict WhisperHipBackend.transcribe_batch returns, rebuilt on every rank from the gathered records (same
    formulas: backend.py transcribe_batch / _align_batch_words, i.e. mlx_lightning.py:82-119, 290-369)."""
    from .audio import N_SAMPLES, SAMPLE_RATE
    from .backend import WhisperHipBackend
    all_segments, by_seg = [], {}
    for rec in records:
        i = rec["chunk_id"]
        seg = segments[i]
        text_ids = [t for t in rec["tokens"] if t < tokenizer.eot]
        text = tokenizer.decode(text_ids).strip()
        by_seg[i] = rec
        if not text:
            continue
        dur = min(len(seg["audio"]), N_SAMPLES) / SAMPLE_RATE
        s = {"start": seg["start"] + 0.0, "end": min(seg["start"] + 0.0 + dur, seg["end"]), "text": text, "id": len(all_segments)}
        if dtw:
            words, a = [], 0
            for tok_end, s_ms, e_ms in rec["word_spans"]:
                w = {"word": tokenizer.decode(text_ids[a:tok_end]).strip(), "start": s_ms / 1000.0, "end": e_ms / 1000.0,
                     "probability": 1.0}
                a = tok_end
                words.append(dict(w, start=w["start"] + s["start"], end=min(w["end"] + s["start"], s["end"])))
            s["words"] = words
        all_segments.append(s)
    result = {"segments": all_segments, "language": language}
    if align_words and segments:
        aligned_segments = []
        for vi, rel in WhisperHipBackend._group_by_vad(all_segments, segments):
            payload = by_seg[vi]["aligned"]
            if payload is None:
                raise RuntimeError(f"segment {vi}: its record carries no alignment (the rank that held it did not align)")
            aligned = assemble_aligned(rel, payload, language)
            aligned_segments += WhisperHipBackend._offset_aligned(aligned, segments[vi]["start"])
        result["segments"] = aligned_segments
    return result
