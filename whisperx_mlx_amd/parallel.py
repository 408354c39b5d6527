"""Multi-GPU sharding of the chunk list.  Every <= 30 s chunk is independent end
to end (the reference assumes the same: whisperx/asr.py:70-87, condition_on_previous_text
False at whisperx/backends/mlx_whisper.py:79), so chunks are dealt to ranks with no
data-path collective and the fixed-width result records come back with ONE all_gather
(RCCL over xGMI when the process group is "nccl"; gloo on CPU for the tests)."""
from typing import Dict, List, Sequence

import numpy as np
import torch

MAX_TOK = 224          # sample_len
REC_W = 4 + MAX_TOK + 1 + 3 * MAX_TOK
# int32 record: [chunk_id, n_tokens, sum_logprob bits, no_speech bits, tokens[224], n_words,
#                word_tok_end[224], word_start_ms[224], word_end_ms[224]]


def shard_indices(durations: Sequence[float], rank: int, world: int) -> List[int]:
    """Longest-first round-robin deal: balances decode length across ranks."""
    order = sorted(range(len(durations)), key=lambda i: (-durations[i], i))
    return sorted(order[rank::world])


def pack_records(results: List[Dict], chunk_ids: Sequence[int]) -> torch.Tensor:
    rec = torch.zeros(len(results), REC_W, dtype=torch.int32)
    for r, (res, cid) in enumerate(zip(results, chunk_ids)):
        toks = list(res["tokens"])[:MAX_TOK]
        rec[r, 0], rec[r, 1] = int(cid), len(toks)
        rec[r, 2] = int(np.float32(res.get("sum_logprob", res.get("avg_logprob", 0.0))).view(np.int32))
        rec[r, 3] = int(np.float32(res.get("no_speech_prob", 0.0)).view(np.int32))
        rec[r, 4: 4 + len(toks)] = torch.tensor(toks, dtype=torch.int32)
        words = res.get("word_spans", [])[:MAX_TOK]      # (tok_end, start_ms, end_ms)
        o = 4 + MAX_TOK
        rec[r, o] = len(words)
        for k, (te, s, e) in enumerate(words):
            rec[r, o + 1 + k] = int(te)
            rec[r, o + 1 + MAX_TOK + k] = int(s)
            rec[r, o + 1 + 2 * MAX_TOK + k] = int(e)
    return rec


def unpack_records(rec: torch.Tensor) -> List[Dict]:
    out = []
    rec = rec.cpu().numpy()
    for row in rec:
        n = int(row[1])
        o = 4 + MAX_TOK
        nw = int(row[o])
        out.append({
            "chunk_id": int(row[0]), "tokens": row[4: 4 + n].tolist(),
            "sum_logprob": float(row[2: 3].view(np.float32)[0]), "no_speech_prob": float(row[3: 4].view(np.float32)[0]),
            "word_spans": [(int(row[o + 1 + k]), int(row[o + 1 + MAX_TOK + k]), int(row[o + 1 + 2 * MAX_TOK + k]))
                           for k in range(nw)]})
    return out


def gather_records(local: torch.Tensor, device=None) -> List[Dict]:
    """One collective: all_gather of the (padded) fixed-width records; every rank returns the
    full list ordered by chunk_id."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return sorted(unpack_records(local), key=lambda r: r["chunk_id"])
    world = dist.get_world_size()
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device())
                                             if dist.get_backend() == "nccl" else torch.device("cpu"))
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)             # sizes (8 bytes per rank), then the payload
    n_max = int(max(c.item() for c in counts))
    pad = torch.full((n_max, REC_W), -1, dtype=torch.int32, device=dev)
    pad[: local.shape[0]] = local.to(dev)
    allr = torch.empty(world * n_max, REC_W, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(allr, pad)
    allr = allr[allr[:, 0] >= 0]
    return sorted(unpack_records(allr), key=lambda r: r["chunk_id"])


def transcribe_sharded(backend, chunks: List[np.ndarray], language="en", task="transcribe", word_timestamps=False):
    """Each rank decodes its shard of `chunks` on its own GPU, then one gather."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    mine = shard_indices([len(c) for c in chunks], rank, world)
    results = backend._decode_chunks([chunks[i] for i in mine], language, task, word_timestamps) if mine else []
    for r in results:
        r["sum_logprob"] = r["avg_logprob"] * (len(r["tokens"]) + 1)
        spans, pos = [], 0
        for w in r.get("words", []):
            spans.append((pos, int(round(w["start"] * 1000)), int(round(w["end"] * 1000))))
        r["word_spans"] = spans
    return gather_records(pack_records(results, mine))
