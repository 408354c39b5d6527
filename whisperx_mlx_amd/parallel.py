"""Multi-GPU sharding of the chunk list.  Every <= 30 s chunk is independent end
to end (the reference assumes the same: whisperx/asr.py:70-87, condition_on_previous_text
False at whisperx/backends/mlx_whisper.py:79), so chunks are dealt to ranks with no
data-path collective and the fixed-width result records come back with ONE all_gather
(RCCL over xGMI when the process group is "nccl"; gloo on CPU for the tests).  Every rank
knows the whole chunk list, hence every rank's share: no size exchange precedes the gather.

What follows the gather costs milliseconds whatever the world size (round 5): the gathered records stay ONE int32 array
(`RecordTable`: columns are numpy slices, a record becomes a dict only when somebody reads it), a rank's own chunks keep
the dicts its launcher threads already built, and `materialize="lazy"` defers the dicts of the other ranks' chunks to the
moment they are read.  Nothing raises between the start of a rank's share and the collective: a rank that fails ships
stub records that say so, and every rank raises the same error AFTER the gather instead of leaving the others inside it."""
import warnings
from collections.abc import Sequence as _SequenceABC
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

MAX_TOK = 224          # sample_len
# int32 record: [chunk_id, n_tokens, sum_logprob bits, no_speech bits, flags, tokens[224], n_words,
#                            word_tok_end[224], word_start_ms[224], word_end_ms[224]               = REC_W_ASR ints
#   -- jobs that force-align (config 4: the alignment stage shards like the chunks do and its result travels in the SAME
#      gather; /root/reference/whisperx/alignment.py:206-373 aligns every segment independently) append --
#                            n_out (-1: no alignment payload), n_aligned_words, n_sentences,
#                            out[48] x (kind 0 failed / 1 aligned, transcript segment, start_ms, end_ms),
#                            sentence[96] x (out index, span begin, span end)   spans index the segment's text,
#                            aligned_word[448] x (start_ms, end_ms, score_milli)   MISSING where align() omits the key]
# Every rank derives the width from the call's own arguments (align_words), like the counts: nothing is exchanged.
HDR = 5
_O_WORDS = HDR + MAX_TOK
REC_W_ASR = _O_WORDS + 1 + 3 * MAX_TOK
_O_ALIGN = REC_W_ASR
MAX_OUT, MAX_SENT = 48, 96
MAX_AWORD = 448        # languages without spaces align one word per character: up to two characters per token
_O_OUT = _O_ALIGN + 3
_O_SENT = _O_OUT + 4 * MAX_OUT
_O_AWORD = _O_SENT + 3 * MAX_SENT
REC_W = _O_AWORD + 3 * MAX_AWORD
MISSING = -2 ** 31
# flags
F_TEXT_KNOWN, F_HAS_TEXT = 1, 2       # the owner says whether the chunk's text is empty (an empty chunk gives no segment)
F_ALIGN_STAGE_FAILED = 4              # the owner's alignment stage raised: the JOB comes back unaligned, as mlx_lightning.py:365-367
F_ALIGN_OVERFLOW = 8                  # this chunk's alignment does not fit the record: the chunk comes back as align()'s failure branch
F_RANK_FAILED = 16                    # the owner raised before the gather; the message travels in the token area


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


def _span_columns(res: Dict):
    """the three word-span columns (tok_end, start_ms, end_ms) of a chunk result as arrays"""
    cols = res.get("word_spans_np")              # int32 arrays straight from the backend's host half (dtw_words.words_upstream)
    if cols is not None and res.get("word_spans") is None:
        return cols
    spans = res.get("word_spans")
    if spans is None:
        words = res.get("words")
        if not words:
            return None
        ends = res.get("word_tok_end")
        if ends is None or len(ends) != len(words):
            spans = word_spans(res)
        else:                                     # the dict form without a Python tuple per word
            n = len(words)
            return (np.asarray(ends, dtype=np.int32),
                    np.rint(np.fromiter((w["start"] for w in words), np.float64, n) * 1000.0).astype(np.int32),
                    np.rint(np.fromiter((w["end"] for w in words), np.float64, n) * 1000.0).astype(np.int32))
    if not spans:
        return None
    w = np.asarray(spans, dtype=np.int32)         # (n_words, 3)
    return w[:, 0], w[:, 1], w[:, 2]


def pack_records(results: List[Dict], chunk_ids: Sequence[int], align: Optional[bool] = None) -> torch.Tensor:
    """per-chunk results -> the fixed-width records.  align=None: the wide record iff a result carries an alignment (a
    job passes its own align_words, so that every rank packs the same width whatever its share holds)."""
    if align is None:
        align = any(r.get("aligned") is not None for r in results)
    width = REC_W if align else REC_W_ASR
    rec = np.zeros((len(results), width), dtype=np.int32)
    fbits = rec[:, 2:4].view(np.float32)          # sum_logprob, no_speech_prob as float32 bit patterns
    for r, (res, cid) in enumerate(zip(results, chunk_ids)):
        toks = res["tokens"]
        if len(toks) > MAX_TOK:
            toks = toks[:MAX_TOK]
        row = rec[r]
        row[0], row[1] = int(cid), len(toks)
        fbits[r, 0], fbits[r, 1] = res.get("sum_logprob", res.get("avg_logprob", 0.0)), res.get("no_speech_prob", 0.0)
        text = res.get("text")
        flags = 0 if text is None else (F_TEXT_KNOWN | (F_HAS_TEXT if text else 0))
        row[HDR: HDR + len(toks)] = toks
        cols = _span_columns(res)
        if cols is not None:
            nw = min(len(cols[0]), MAX_TOK)
            row[_O_WORDS] = nw
            for k in range(3):
                row[_O_WORDS + 1 + k * MAX_TOK: _O_WORDS + 1 + k * MAX_TOK + nw] = cols[k][:nw]
        if align:
            row[_O_ALIGN] = -1
            if res.get("align_failed"):
                flags |= F_ALIGN_STAGE_FAILED
            elif res.get("aligned") is not None:
                try:
                    pack_aligned(row, *res["aligned"])
                except ValueError as e:           # never raise before the collective: the chunk falls back, the job goes on
                    warnings.warn(f"chunk {int(cid)}: {e}; its segments come back unaligned")
                    row[_O_ALIGN:] = 0
                    row[_O_ALIGN] = -1
                    flags |= F_ALIGN_OVERFLOW
        row[4] = flags
    return torch.from_numpy(rec)


def stub_records(chunk_ids: Sequence[int], error: BaseException, align: bool) -> torch.Tensor:
    """what a rank that raised contributes to the gather: its chunk ids, the failure flag and the message"""
    rec = np.zeros((len(chunk_ids), REC_W if align else REC_W_ASR), dtype=np.int32)
    msg = f"{type(error).__name__}: {error}".encode("utf-8", "replace")[: 4 * MAX_TOK - 4]
    for r, cid in enumerate(chunk_ids):
        rec[r, 0], rec[r, 4] = int(cid), F_RANK_FAILED
        if r == 0:
            rec[r, HDR] = len(msg)
            rec[r, HDR + 1: HDR + 1 + (len(msg) + 3) // 4] = np.frombuffer(msg + b"\0" * (-len(msg) % 4), dtype=np.int32)
        if align:
            rec[r, _O_ALIGN] = -1
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
    and the sentence spans (alignment.sentence_word_texts).  ValueError when the result does not fit the block
    (pack_records turns that into the chunk's fallback)."""
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
            if n_w >= MAX_AWORD:
                raise ValueError(f"more than {MAX_AWORD} aligned words in one chunk")
            row[_O_AWORD + 3 * n_w: _O_AWORD + 3 * n_w + 3] = (_ms(w["start"]) if "start" in w else MISSING,
                                                               _ms(w["end"]) if "end" in w else MISSING,
                                                               _ms(w["score"]) if "score" in w else MISSING)
            n_w += 1
    row[_O_ALIGN: _O_ALIGN + 3] = (len(segs), n_w, n_sent)


def unpack_aligned(row: np.ndarray) -> Optional[Dict]:
    if row.shape[0] <= REC_W_ASR:
        return None
    n_out = int(row[_O_ALIGN])
    if n_out < 0:
        return None
    outs = row[_O_OUT: _O_OUT + 4 * n_out].reshape(n_out, 4).tolist()
    out = [tuple(o) + ([],) for o in outs]
    n_sent = int(row[_O_ALIGN + 2])
    for o, sb, se in row[_O_SENT: _O_SENT + 3 * n_sent].reshape(n_sent, 3).tolist():
        out[o][4].append((sb, se))
    n_w = int(row[_O_ALIGN + 1])
    words = [tuple(w) for w in row[_O_AWORD: _O_AWORD + 3 * n_w].reshape(n_w, 3).tolist()]
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


def record_dict(row: np.ndarray) -> Dict:
    """one record as the dict the round-1..4 callers read"""
    n, nw = int(row[1]), int(row[_O_WORDS])
    f = row[2:4].view(np.float32)
    cols = [row[_O_WORDS + 1 + k * MAX_TOK: _O_WORDS + 1 + k * MAX_TOK + nw].tolist() for k in range(3)]
    return {"chunk_id": int(row[0]), "tokens": row[HDR: HDR + n].tolist(), "sum_logprob": float(f[0]),
            "no_speech_prob": float(f[1]), "word_spans": list(zip(*cols)) if nw else [],
            "aligned": unpack_aligned(row)}


class RecordTable(_SequenceABC):
    """The records of a job as ONE int32 array (+ the order that sorts them by chunk id): what `gather_records` hands
    back.  Columns are numpy views -- `chunk_ids`, `n_tokens`, `sum_logprob`, `flags`, `tokens_of(k)` -- and `table[k]`
    builds the k-th record's dict when somebody asks for it; a table compares equal to the list of those dicts.
    Unpacking 2 560 records (8 ranks x the driver's 320 chunks) into dicts took 446 ms of interpreter time on every
    rank; building the table takes the argsort of 2 560 ids."""

    def __init__(self, rec: np.ndarray, order: Optional[np.ndarray] = None):
        self.rec = rec
        self.order = np.arange(rec.shape[0]) if order is None else order
        self.align = rec.shape[1] > REC_W_ASR

    def __len__(self):
        return int(self.order.shape[0])

    def __getitem__(self, k):
        if isinstance(k, slice):
            return RecordTable(self.rec, self.order[k])
        return record_dict(self.rec[self.order[k]])

    def __iter__(self):
        for j in self.order.tolist():
            yield record_dict(self.rec[j])

    def __eq__(self, other):
        if isinstance(other, (RecordTable, list, tuple)):
            return len(self) == len(other) and all(a == b for a, b in zip(self, other))
        return NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = None

    def row(self, k) -> np.ndarray:
        return self.rec[self.order[k]]

    def col(self, c) -> np.ndarray:
        return self.rec[self.order, c]

    @property
    def chunk_ids(self) -> np.ndarray:
        return self.col(0)

    @property
    def n_tokens(self) -> np.ndarray:
        return self.col(1)

    @property
    def sum_logprob(self) -> np.ndarray:
        return self.col(2).view(np.float32)

    @property
    def no_speech_prob(self) -> np.ndarray:
        return self.col(3).view(np.float32)

    @property
    def flags(self) -> np.ndarray:
        return self.col(4)

    def tokens_of(self, k) -> np.ndarray:
        row = self.row(k)
        return row[HDR: HDR + int(row[1])]

    def failure(self) -> Optional[str]:
        """the message of the first rank that shipped stub records, or None"""
        bad = np.nonzero(self.flags & F_RANK_FAILED)[0]
        if bad.size == 0:
            return None
        for k in bad.tolist():
            row = self.row(k)
            n = int(row[HDR])
            if n > 0:
                return (f"the rank that held chunk {int(row[0])} failed before the gather: "
                        + row[HDR + 1: HDR + 1 + (n + 3) // 4].tobytes()[:n].decode("utf-8", "replace"))
        return f"the rank that held chunk {int(self.row(int(bad[0]))[0])} failed before the gather"


def unpack_records(rec) -> RecordTable:
    """records (tensor or array) -> a RecordTable in the order given"""
    if torch.is_tensor(rec):
        rec = rec.cpu().numpy()
    return RecordTable(np.ascontiguousarray(rec, dtype=np.int32))


def _sorted_table(rec: np.ndarray, valid: Optional[np.ndarray] = None) -> RecordTable:
    idx = np.arange(rec.shape[0]) if valid is None else valid
    return RecordTable(rec, idx[np.argsort(rec[idx, 0], kind="stable")])


def gather_records(local: torch.Tensor, counts: Optional[Sequence[int]] = None, device=None) -> RecordTable:
    """THE collective: one all_gather of the fixed-width records, padded to the largest share; every rank returns
    the full table ordered by chunk_id.  `counts[r]` = records rank r contributes (computable on every rank from
    shard_indices, so no size exchange).  counts=None (a caller that does not know the other ranks' shares): the
    shares are exchanged first with one 8-byte all_gather -- a second collective, so the product paths always pass
    `counts`."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return _sorted_table(unpack_records(local).rec)
    world = dist.get_world_size()
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device())
                                             if dist.get_backend() == "nccl" else torch.device("cpu"))
    if counts is None:
        sizes = torch.zeros(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(sizes, torch.tensor([local.shape[0]], dtype=torch.int64, device=dev))
        counts = [int(v) for v in sizes.tolist()]
    n_max = int(max(counts))
    width = int(local.shape[1])
    assert local.shape[0] == counts[dist.get_rank()], "gather_records: counts[rank] must be this rank's number of records"
    if local.shape[0] == n_max:
        pad = local.to(dev).contiguous()
    else:
        pad = torch.zeros((n_max, width), dtype=torch.int32, device=dev)
        pad[: local.shape[0]] = local.to(dev)
    allr = torch.empty(world * n_max, width, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(allr, pad)
    valid = np.concatenate([r * n_max + np.arange(int(c)) for r, c in enumerate(counts)]) if world else np.zeros(0, np.int64)
    return _sorted_table(allr.cpu().numpy(), valid)       # rows past a rank's count are padding: never looked at


def transcribe_sharded(backend, chunks: List[np.ndarray], language=None, task="transcribe", word_timestamps=False):
    """Each rank decodes its shard of `chunks` on its own GPU, then one gather.  With language=None the language is
    detected ONCE, from the first chunk of the whole list (every rank holds it and runs the same deterministic
    kernels), so all ranks decode with the same prompt.  Returns the RecordTable of the whole job on every rank."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lens = [len(c) for c in chunks]
    shares = [shard_indices(lens, r, world) for r in range(world)]
    mine = shares[rank]
    err = None
    try:
        if language is None:
            language = backend.detect_language(chunks[0]) if (chunks and backend.is_multilingual) else "en"
        results = backend._decode_chunks([chunks[i] for i in mine], language, task, word_timestamps) if mine else []
        for r in results:
            r.setdefault("sum_logprob", r["avg_logprob"] * (len(r["tokens"]) + 1))
        local = pack_records(results, mine, align=False)
    except Exception as e:                       # noqa: BLE001 - raised on every rank after the gather
        err = e
        local = stub_records(mine, e, align=False)
    table = gather_records(local, counts=[len(s) for s in shares])
    if err is not None:
        raise err
    if table.failure():
        raise RuntimeError(table.failure())
    return table


def transcribe_batch_sharded(backend, segments: List[Dict], batch_size: int = 16, align_words: bool = False, language=None,
                             task: str = "transcribe", word_timestamps=False, materialize: str = "all",
                             return_records: bool = False, reuse_own: bool = True, timings: Optional[Dict] = None, **kw) -> Dict:
    """`WhisperHipBackend.transcribe_batch` over the GPUs of one node (BASELINE.json config 4 as written: large-v3 +
    wav2vec2 align_model, VAD chunks sharded over the ranks, one RCCL gather): every rank transcribes -- and, with
    align_words, force-aligns (its own W2VHipModel) -- its share of the VAD segments, the fixed-width records (token ids,
    log-probabilities, DTW word spans, aligned words) come back with ONE all_gather, and every rank returns the dict
    the single-process call returns.

    segments: [{"start", "end", "audio"}] as whisperx/asr.py:70-73 builds them from the VAD's merged chunks, each <= 30 s
    (Vad.merge_chunks guarantees it for chunk_size 30; one segment = one chunk = one record).  The reference aligns
    every segment independently (/root/reference/whisperx/alignment.py:206-373, driven per VAD segment by
    /root/reference/whisperx/backends/mlx_lightning.py:290-369), so the alignment stage shards exactly like the ASR.

    materialize: "all" -- result["segments"] is a plain list on every rank (the rank's own chunks keep the dicts they
    already have; the other ranks' are decoded from their records: O(job) interpreter time per rank); "lazy" -- a
    `LazySegments` sequence of the same length that equals that list and builds a chunk's dicts when they are read;
    "root" -- the list on rank 0, lazy elsewhere.  return_records=True adds result["records"], the job's RecordTable.
    reuse_own=False rebuilds the rank's own chunks from their records too (tests: pack -> unpack -> assemble for every chunk).
    timings: a dict that receives the seconds of this rank's share ("local"), packing ("pack"), the collective ("gather")
    and table + result ("assemble").

    Failures: nothing raises between a rank's first kernel and the gather.  A rank whose share fails ships stub
    records and EVERY rank raises the same RuntimeError after the collective.  An alignment stage that fails on any
    rank gives the whole job back unaligned with a warning, as the single-process call does (mlx_lightning.py:365-367)."""
    import time
    import torch.distributed as dist
    from .audio import N_SAMPLES
    t_start = time.perf_counter()
    if materialize not in ("all", "lazy", "root"):
        raise ValueError(f"materialize={materialize!r}")
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    has_audio = [s.get("audio") is not None for s in segments]           # (segments without audio are skipped, as transcribe_batch does)
    lens = [int(len(s["audio"])) if ok else 0 for s, ok in zip(segments, has_audio)]
    if any(n > N_SAMPLES for n in lens):              # (the same on every rank, before anybody has started)
        raise ValueError("transcribe_batch_sharded: segments must be <= 30 s (one chunk each); split longer ones first")
    shares = [shard_indices(lens, r, world) for r in range(world)]
    mine = [i for i in shares[rank] if has_audio[i]]
    counts = [sum(1 for i in sh if has_audio[i]) for sh in shares]
    my_segments = [segments[i] for i in mine]
    dtw = word_timestamps if word_timestamps in ("dtw", "dtw_inrepo") else (word_timestamps is True and not align_words)
    chunks: List[Dict] = []
    err = None
    try:
        if language is None:
            first = next((s["audio"] for s, ok in zip(segments, has_audio) if ok), None)
            language = backend.detect_language(first) if (first is not None and backend.is_multilingual) else "en"
        if kw.get("forced_lens") is not None:            # bench workload: one forced length per segment of the WHOLE list
            kw = dict(kw, forced_lens=[kw["forced_lens"][i] for i in mine])
        if my_segments:
            res = backend.transcribe_batch(my_segments, batch_size=batch_size, language=language, task=task, word_timestamps=dtw,
                                           return_chunks=True, align_words=False, **kw)
            chunks = res["chunks"]
            assert [c["segment"] for c in chunks] == list(range(len(my_segments)))
            if align_words:
                try:
                    groups = backend._group_by_vad(res["segments"], my_segments)
                    trace: List[list] = []
                    aligned = backend.align_groups(groups, my_segments, language, _trace=trace)
                    for (vi, _rel), a, tr in zip(groups, aligned, trace):
                        chunks[vi]["aligned"] = (a, tr)
                except Exception as e:            # noqa: BLE001 - mlx_lightning.py:365-367: a warning and the unaligned result
                    print(f"Warning: Batch word alignment failed: {e}")
                    for c in chunks:
                        c.pop("aligned", None)
                        c["align_failed"] = True
        t_local = time.perf_counter()
        for c in chunks:
            c.setdefault("sum_logprob", c["avg_logprob"] * (len(c["tokens"]) + 1))
        local = pack_records(chunks, mine, align=bool(align_words))
    except Exception as e:                       # noqa: BLE001 - raised on every rank after the gather
        err = e
        t_local = time.perf_counter()
        local = stub_records(mine, e, align=bool(align_words))
    t_pack = time.perf_counter()
    table = gather_records(local, counts=counts)
    t_gather = time.perf_counter()
    if err is not None:
        raise err
    if table.failure():
        raise RuntimeError(table.failure())
    own = {i: c for i, c in zip(mine, chunks)} if reuse_own else {}
    lazy = materialize == "lazy" or (materialize == "root" and rank != 0)
    result = assemble_result(backend.tokenizer, segments, table, language, dtw=dtw, align_words=align_words, own=own, lazy=lazy)
    if return_records:
        result["records"] = table
    if timings is not None:
        timings.update(local=t_local - t_start, pack=t_pack - t_local, gather=t_gather - t_pack,
                       assemble=time.perf_counter() - t_gather)
    return result


class LazySegments(_SequenceABC):
    """result["segments"] of a sharded job whose dicts are built per chunk when they are read (and kept).  Same length,
    same items, same order as the list `materialize="all"` returns; `list(x)` is that list."""

    def __init__(self, first: np.ndarray, build):
        self._first, self._build, self._done = first, build, {}

    def __len__(self):
        return int(self._first[-1])

    def _chunk(self, k):
        got = self._done.get(k)
        if got is None:
            got = self._done[k] = self._build(k)
            assert len(got) == int(self._first[k + 1] - self._first[k]), "a chunk's record and its segments disagree"
        return got

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        k = int(np.searchsorted(self._first, i, side="right")) - 1
        return self._chunk(k)[i - int(self._first[k])]

    def __iter__(self):
        for k in np.nonzero(np.diff(self._first))[0].tolist():
            yield from self._chunk(k)

    def __eq__(self, other):
        if isinstance(other, (LazySegments, list, tuple)):
            return len(self) == len(other) and all(a == b for a, b in zip(self, other))
        return NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = None


def _word_text(tokenizer, ids) -> str:
    """a DTW word's text: its tokens decoded ONE BY ONE and joined (dtw_words.words_upstream via split_to_words;
    /root/reference/mlx_whisper_optimized_final.py:210) -- a joint decode differs when a multi-byte character is split
    over tokens"""
    return "".join(tokenizer.decode_token(t) for t in ids).strip()


def assemble_result(tokenizer, segments: List[Dict], records, language: str, dtw=False, align_words: bool = False,
                    own: Optional[Dict[int, Dict]] = None, lazy: bool = False) -> Dict:
    """the dict WhisperHipBackend.transcribe_batch returns, rebuilt from the gathered records (same formulas: backend.py
    transcribe_batch / _align_batch_words, i.e. mlx_lightning.py:82-119, 290-369).  `own`: the per-chunk results this
    rank built itself, by chunk id -- their text, words and align() dicts are taken as they are.  What happens up front
    is column arithmetic over the table (which chunks have text, how many segments each contributes, segment ids); a
    chunk's dicts are built by `chunk(k)`, for every chunk now (lazy=False) or when read."""
    from .audio import N_SAMPLES, SAMPLE_RATE
    from .backend import WhisperHipBackend
    table = records if isinstance(records, RecordTable) else _table_from_dicts(records, align_words)
    own = own or {}
    n = len(table)
    ids = table.chunk_ids.tolist()
    flags = table.flags
    eot = tokenizer.eot

    def text_ids_of(k):
        t = table.tokens_of(k)
        return t[t < eot].tolist()

    has_text = (flags & F_HAS_TEXT) != 0
    for k in np.nonzero((flags & F_TEXT_KNOWN) == 0)[0].tolist():       # records packed without their text: decode to know
        has_text[k] = bool(tokenizer.decode(text_ids_of(k)).strip())
    seg_id = np.cumsum(has_text) - 1
    unaligned_job = False
    if align_words:
        if np.any(flags & F_ALIGN_STAGE_FAILED):
            print("Warning: Batch word alignment failed on a rank of the job")
            print("Returning transcription without word-level timestamps")
            unaligned_job = True
        else:
            n_out = table.col(_O_ALIGN) if table.align else np.full(n, -1)
            overflow = (flags & F_ALIGN_OVERFLOW) != 0
            bad = has_text & (n_out < 0) & ~overflow
            if np.any(bad):
                raise RuntimeError(f"segment {ids[int(np.nonzero(bad)[0][0])]}: its record carries no alignment (the rank that held it did not align)")
    if align_words and not unaligned_job:
        per_chunk = np.where(has_text, np.where(overflow, 1, np.maximum(n_out, 0)), 0)
    else:
        per_chunk = has_text.astype(np.int64)
    first = np.concatenate([[0], np.cumsum(per_chunk)]).astype(np.int64)

    def asr_segment(k):
        i = ids[k]
        seg, mine = segments[i], own.get(i)
        text = mine["text"] if (mine is not None and "text" in mine) else tokenizer.decode(text_ids_of(k)).strip()
        dur = min(len(seg["audio"]), N_SAMPLES) / SAMPLE_RATE
        s = {"start": seg["start"] + 0.0, "end": min(seg["start"] + 0.0 + dur, seg["end"]), "text": text, "id": int(seg_id[k])}
        if dtw:
            if mine is not None and "words" in mine:
                s["words"] = [dict(w, start=w["start"] + s["start"], end=min(w["end"] + s["start"], s["end"])) for w in mine["words"]]
            else:
                # the same doubles as above, column-wise: w["start"] = ms / 1000.0 (a frame index / 50 on the owner: the same
                # rational, so the same double), then + s["start"] and min(., s["end"]) in float64
                row, tid = table.row(k), text_ids_of(k)
                nw = int(row[_O_WORDS])
                cols = row[_O_WORDS + 1: _O_WORDS + 1 + 3 * MAX_TOK].reshape(3, MAX_TOK)[:, :nw]
                st = (cols[1] / 1000.0 + s["start"]).tolist()
                en = np.minimum(cols[2] / 1000.0 + s["start"], s["end"]).tolist()
                texts = [tokenizer.decode_token(t) for t in tid]
                ends = cols[0].tolist()
                s["words"] = [{"word": "".join(texts[a:b]).strip(), "start": s0, "end": e0, "probability": 1.0}
                              for a, b, s0, e0 in zip([0] + ends[:-1], ends, st, en)]
        return s

    def chunk(k):
        if not has_text[k]:
            return []
        s = asr_segment(k)
        if not align_words or unaligned_job:
            return [s]
        i = ids[k]
        vad = segments[i]
        rel = dict(s)
        rel["start"] -= vad["start"]
        rel["end"] -= vad["start"]
        mine = own.get(i)
        if flags[k] & F_ALIGN_OVERFLOW:
            aligned = {"segments": [{"start": rel["start"], "end": rel["end"], "text": rel["text"], "words": [], "chars": None}]}
        elif mine is not None and mine.get("aligned") is not None:
            aligned = mine["aligned"][0]
        else:
            aligned = assemble_aligned([rel], unpack_aligned(table.row(k)), language)
        return WhisperHipBackend._offset_aligned(aligned, vad["start"])

    result = {"language": language}
    if lazy:
        result["segments"] = LazySegments(first, chunk)
    else:
        result["segments"] = [s for k in range(n) for s in chunk(k)]
        assert len(result["segments"]) == int(first[-1])
    return result


def _table_from_dicts(records: List[Dict], align: bool) -> RecordTable:
    """record dicts (round-4 callers of assemble_result) -> a table"""
    rows = []
    for r in records:
        d = dict(r)
        if d.get("aligned") is not None and not isinstance(d["aligned"], tuple):
            raise TypeError("assemble_result takes the RecordTable gather_records returns")
        d.pop("aligned", None)
        rows.append(d)
    return _sorted_table(pack_records(rows, [r["chunk_id"] for r in records], align=align).numpy())
