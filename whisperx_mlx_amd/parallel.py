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
REC_W = 4 + MAX_TOK + 1 + 3 * MAX_TOK
# int32 record: [chunk_id, n_tokens, sum_logprob bits, no_speech bits, tokens[224], n_words,
#                            word_tok_end[224], word_start_ms[224], word_end_ms[224]]
_O_WORDS = 4 + MAX_TOK


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
    for r, (res, cid) in enumerate(zip(results, chunk_ids)):
        toks = np.asarray(list(res["tokens"])[:MAX_TOK], dtype=np.int32)
        row = rec[r]
        row[0], row[1] = int(cid), len(toks)
        row[2:4] = np.array([res.get("sum_logprob", res.get("avg_logprob", 0.0)), res.get("no_speech_prob", 0.0)],
                            dtype=np.float32).view(np.int32)
        row[4: 4 + len(toks)] = toks
        words = res.get("word_spans")
        if words is None:
            words = word_spans(res)
        words = words[:MAX_TOK]
        row[_O_WORDS] = len(words)
        if words:
            w = np.asarray(words, dtype=np.int32)         # (n_words, 3): tok_end, start_ms, end_ms
            for k in range(3):
                row[_O_WORDS + 1 + k * MAX_TOK: _O_WORDS + 1 + k * MAX_TOK + len(words)] = w[:, k]
    return torch.from_numpy(rec)


def unpack_records(rec: torch.Tensor) -> List[Dict]:
    out = []
    rec = rec.cpu().numpy()
    for row in rec:
        n = int(row[1])
        nw = int(row[_O_WORDS])
        f = row[2:4].view(np.float32)
        cols = [row[_O_WORDS + 1 + k * MAX_TOK: _O_WORDS + 1 + k * MAX_TOK + nw].tolist() for k in range(3)]
        out.append({"chunk_id": int(row[0]), "tokens": row[4: 4 + n].tolist(), "sum_logprob": float(f[0]),
                    "no_speech_prob": float(f[1]), "word_spans": list(zip(*cols)) if nw else []})
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
