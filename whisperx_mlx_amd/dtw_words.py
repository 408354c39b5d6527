"""Word times from a cross-attention DTW path (host bookkeeping behind wx_dtw_path).

Two variants, as the reference has them: the published `find_alignment` bookkeeping (the default) and the reference's own
in-repo one (/root/reference/mlx_whisper_optimized_final.py:215-251), selected with dtw_variant="inrepo" /
word_timestamps="dtw_inrepo".  Both take (n_rows, path) with path a (2, L) int32 array in start -> end order: row 0 the
token index, row 1 the frame index of every step of the path.
"""
import numpy as np

from .audio import TOKENS_PER_SECOND

_NO_SPANS = (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32))


def words_upstream(tokenizer, text_ids, path_info, spans=None):
    """word times from the DTW path over the alignment matrix rows (text tokens + EOT): published
    find_alignment bookkeeping -- a jump of the token index along the path marks the first frame of a token,
    word k starts at the jump of its first token and ends at the jump of the next word's first token (the EOT row
    for the last word), 20 ms per frame."""
    n_rows, path = path_info
    if n_rows < 2 or path.shape[1] == 0 or not text_ids:
        if spans is not None:
            spans.append(_NO_SPANS)
        return []
    ti, fi = path[0], path[1]
    jumps = np.concatenate([[True], np.diff(ti) > 0])
    jump_times = fi[jumps].astype(np.float64) / TOKENS_PER_SECOND
    words, bounds = tokenizer.split_to_words(text_ids)
    # (vectorised: this runs for every word of every chunk on the launcher threads, after the GPU has finished)
    nj = len(jump_times)
    a, b = bounds[:-1], bounds[1:]
    keep = int(np.searchsorted(a, nj))                 # words whose first token has a jump (a is increasing)
    starts = jump_times[a[:keep]]
    ends = np.maximum(jump_times[np.minimum(b[:keep], nj - 1)], starts)
    out = [{"word": ws, "start": s0, "end": e0, "probability": 1.0, "tok_end": be}
           for ws, s0, e0, be in zip((w.strip() for w in words[:keep]), starts.tolist(), ends.tolist(), b[:keep].tolist()) if ws]
    if spans is not None:
        # the same words as the fixed-width record carries them (parallel.pack_records): token end, start and end in ms.
        # A jump time is a frame index / 50: whole multiples of 20 ms, so the rounding is exact
        m = np.fromiter((bool(w.strip()) for w in words[:keep]), dtype=bool, count=keep)
        spans.append((b[:keep][m].astype(np.int32), np.rint(starts[m] * 1000.0).astype(np.int32),
                      np.rint(ends[m] * 1000.0).astype(np.int32)))
    return out


def words_inrepo(tokenizer, text_ids, path_info):
    """the reference's own in-repo bookkeeping (mlx_whisper_optimized_final.py:215-251): row 0 of dtw(-W.T) (frame
    indices along the path) is looked up by TOKEN index -- a word starts at row0[first token], ends at row0[last token]
    (the last word at row0[-1]), 20 ms per frame, end >= start.  Kept as the reference has it, including that the path
    position, not the token's row, is what the index selects."""
    _n_rows, path = path_info
    if path.shape[1] == 0 or not text_ids:
        return []
    row0 = path[0]
    n = len(row0)
    words, word_tokens = tokenizer.split_to_word_tokens(text_ids)
    bounds = np.concatenate([[0], np.cumsum([len(t) for t in word_tokens])])
    res = []
    for k, (w, a, b) in enumerate(zip(words, bounds[:-1], bounds[1:])):
        if not w.strip():
            continue
        last = k == len(words) - 1
        if last and a >= n:
            continue
        f0 = int(row0[a]) if a < n else 0
        f1 = int(row0[-1]) if last else (int(row0[b - 1]) if b - 1 < n else f0)
        res.append({"word": w.strip(), "start": float(f0 * 0.02), "end": float(max(f1, f0) * 0.02), "probability": 1.0,
                    "tok_end": int(b)})
    return res
