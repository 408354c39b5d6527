"""Oracle: cross-attention DTW word timing.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED at the reference boundary: ``dtw`` is imported from third-party
``mlx_whisper.timing`` (/root/reference/mlx_whisper_optimized_final.py:24, called
:201), which is not vendored.  This file restates the published OpenAI-Whisper
algorithm that mlx-whisper re-implements (whisper/timing.py ``dtw_cpu`` /
``backtrace`` / ``median_filter`` / ``find_alignment``) and the in-repo variant
``extract_words_with_dtw`` (mlx_whisper_optimized_final.py:128-253) with its
median filter (median_filter_fix.py:6-21).  Cross-checked in tests against
``transformers.models.whisper.generation_whisper._dynamic_time_warping`` /
``_median_filter`` (third-party secondary oracle).
"""
import numpy as np

F32 = np.float32


def dtw_path(x):
    """Classic DTW on cost matrix x (N, M) f32.  Published algorithm (dtw_cpu):
    cost[i,j] = x[i-1,j-1] + min(c0=diag, c1=up(i-1,j), c2=left(i,j-1)) with the
    strict-< tie rule (diag only if strictly smallest, else up only if strictly
    smallest, else left); columns outer, rows inner.  Returns int32 (2, L):
    row 0 = i indices, row 1 = j indices, from (0,0) to (N-1,M-1)."""
    x = np.asarray(x, dtype=F32)
    N, M = x.shape
    cost = np.full((N + 1, M + 1), np.inf, dtype=F32)
    trace = -np.ones((N + 1, M + 1), dtype=np.int8)
    cost[0, 0] = 0
    for j in range(1, M + 1):
        for i in range(1, N + 1):
            c0 = cost[i - 1, j - 1]
            c1 = cost[i - 1, j]
            c2 = cost[i, j - 1]
            if c0 < c1 and c0 < c2:
                c, t = c0, 0
            elif c1 < c0 and c1 < c2:
                c, t = c1, 1
            else:
                c, t = c2, 2
            cost[i, j] = F32(x[i - 1, j - 1] + c)
            trace[i, j] = t
    return backtrace(trace)


def backtrace(trace):
    i = trace.shape[0] - 1
    j = trace.shape[1] - 1
    trace = trace.copy()
    trace[0, :] = 2
    trace[:, 0] = 1
    res = []
    while i > 0 or j > 0:
        res.append((i - 1, j - 1))
        if trace[i, j] == 0:
            i -= 1
            j -= 1
        elif trace[i, j] == 1:
            i -= 1
        elif trace[i, j] == 2:
            j -= 1
        else:
            raise ValueError("bad trace")
    res = np.array(res, dtype=np.int32).reshape(-1, 2)
    return res[::-1, :].T.copy()


def dtw_path_fast(x):
    """Same result as dtw_path, anti-diagonal free numpy row sweep for big inputs
    (used at full sizes where the pure-python loop is too slow)."""
    x = np.asarray(x, dtype=F32)
    N, M = x.shape
    cost = np.full((N + 1, M + 1), np.inf, dtype=F32)
    trace = -np.ones((N + 1, M + 1), dtype=np.int8)
    cost[0, 0] = 0
    # column-outer order == any order that respects dependencies; do row-outer
    # with a python inner loop only over the sequential 'left' dependency.
    for i in range(1, N + 1):
        up_d = cost[i - 1]
        row = cost[i]
        xr = x[i - 1]
        tr = trace[i]
        for j in range(1, M + 1):
            c0 = up_d[j - 1]
            c1 = up_d[j]
            c2 = row[j - 1]
            if c0 < c1 and c0 < c2:
                c, t = c0, 0
            elif c1 < c0 and c1 < c2:
                c, t = c1, 1
            else:
                c, t = c2, 2
            row[j] = F32(xr[j - 1] + c)
            tr[j] = t
    return backtrace(trace)


def median_filter_reflect(x, width=7):
    """median_filter_fix.py:6-21 (== upstream median_filter): reflect-pad
    width//2 along the last axis, running median of `width`."""
    x = np.asarray(x, dtype=F32)
    pad = width // 2
    if x.shape[-1] <= pad:
        return x
    xp = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(pad, pad)], mode="reflect")
    win = np.lib.stride_tricks.sliding_window_view(xp, width, axis=-1)
    return np.sort(win, axis=-1)[..., pad].astype(F32)


def softmax_f32(x, axis=-1):
    x = np.asarray(x, dtype=F32)
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis=axis, keepdims=True)).astype(F32)


def alignment_matrix_upstream(qk, qk_scale=1.0, medfilt_width=7):
    """Published find_alignment: qk (heads, n_tok, frames) pre-softmax f32 ->
    softmax over frames, z-norm over the TOKEN axis (population std), median-7
    over frames, mean over heads.  Returns (n_tok, frames) f32."""
    w = softmax_f32(np.asarray(qk, dtype=F32) * F32(qk_scale), axis=-1)
    mean = w.mean(axis=-2, keepdims=True, dtype=F32)
    std = np.sqrt(((w - mean) ** 2).mean(axis=-2, keepdims=True, dtype=F32))
    w = (w - mean) / std
    w = median_filter_reflect(w, medfilt_width)
    return w.mean(axis=0, dtype=F32).astype(F32)


def alignment_matrix_inrepo(qk):
    """mlx_whisper_optimized_final.py:171-197: mean over heads FIRST (:171), then
    softmax(W*10) over frames (:185-186), median-7 (:192), per-row z-norm with
    std + 1e-8 (:195-197).  qk (heads, n_tok, frames)."""
    w = np.asarray(qk, dtype=F32).mean(axis=0, dtype=F32)
    w = softmax_f32(w * F32(10.0), axis=-1)
    w = median_filter_reflect(w, 7)
    mean = w.mean(axis=1, keepdims=True, dtype=F32)
    std = w.std(axis=1, keepdims=True, dtype=F32) + F32(1e-8)
    return ((w - mean) / std).astype(F32)


def token_times_upstream(matrix):
    """find_alignment tail: text_indices,time_indices = dtw(-matrix); jumps of
    text index -> first frame of every token row.  Returns int32 (n_tok,) frame
    index at which each token row begins (jump_times * 50)."""
    ti, fi = dtw_path_fast(-np.asarray(matrix, dtype=F32))
    jumps = np.concatenate([[1], np.diff(ti)]).astype(bool)
    return fi[jumps].astype(np.int32)


def word_times_upstream(matrix, word_token_counts):
    """start/end seconds per word: boundaries = cumsum of token counts
    (find_alignment: word_boundaries / jump_times[...]); matrix has one extra
    trailing row for EOT.  20 ms per frame."""
    jump_frames = token_times_upstream(matrix)
    b = np.concatenate([[0], np.cumsum(word_token_counts)]).astype(np.int64)
    jt = jump_frames.astype(np.float64) / 50.0
    return jt[b[:-1]], jt[b[1:]]


def inrepo_row0(matrix):
    """mlx_whisper_optimized_final.py:201: alignment = dtw(-W.T); the in-repo
    variant indexes row 0 of that result by token index (:221-233)."""
    return dtw_path_fast(-np.asarray(matrix, dtype=F32).T)[0]


def word_times_inrepo(row0, token_strs):
    """mlx_whisper_optimized_final.py:215-251: group token strings into words (a token that starts with a space
    opens a new word, :222), index row 0 of dtw(-W.T) by TOKEN index: start = row0[first token of the word] (:226),
    end = row0[last token of the word] (:227) clamped to >= start (:230), 20 ms per frame (:235-236); the last word
    ends at row0[-1] (:245).  Returns [(word, start_s, end_s)]; whitespace-only words are dropped (:224, :243)."""
    n = len(row0)
    out = []
    cur, w0 = "", 0
    for i, s in enumerate(token_strs):
        if i > 0 and s.startswith(" "):
            if cur.strip():
                a = int(row0[w0]) if w0 < n else 0
                b = int(row0[i - 1]) if i - 1 < n else a
                out.append((cur.strip(), float(a * 0.02), float(max(b, a) * 0.02)))
            cur, w0 = s, i
        else:
            cur += s
    if cur.strip() and w0 < n:
        a = int(row0[w0])
        b = int(row0[-1]) if n > 0 else a
        out.append((cur.strip(), float(a * 0.02), float(max(b, a) * 0.02)))
    return out
