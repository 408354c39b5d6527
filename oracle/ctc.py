"""Oracle: CTC forced-alignment DP.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

numpy restatement of /root/reference/whisperx/alignment.py:
  get_trellis :387-404, get_wildcard_emission :407-437, backtrack_beam :500-579,
  merge_repeats :597-613, and the seconds mapping :278-290.
All arithmetic is float32 in the same order as the torch code so the trellis
is bit-exact.  Pinned by tests/golden/ctc_*.npz (tools/make_golden.py runs the
reference's own functions).
"""
import math
import numpy as np

F32 = np.float32
NEG_INF = F32(-np.inf)
POS_INF = F32(np.inf)


def wildcard_max(frame_emission, blank_id):
    """max over non-blank labels of one frame -- alignment.py:430-432."""
    e = np.array(frame_emission, dtype=F32, copy=True)
    e[blank_id] = NEG_INF
    return e.max()


def token_emission(frame_emission, tokens, blank_id):
    """alignment.py:407-437: emission of each token, wildcard (-1) -> max non-blank."""
    tokens = np.asarray(tokens, dtype=np.int64)
    reg = frame_emission[np.clip(tokens, 0, None)]
    return np.where(tokens == -1, wildcard_max(frame_emission, blank_id), reg).astype(F32)


def get_trellis(emission, tokens, blank_id=0):
    """alignment.py:387-404.  emission (T,V) f32 log-probs; tokens list[int] (N)."""
    emission = np.asarray(emission, dtype=F32)
    T = emission.shape[0]
    N = len(tokens)
    trellis = np.zeros((T, N), dtype=F32)
    # torch.cumsum on CPU accumulates f32 input in a float64 running sum and
    # rounds every prefix to f32 (verified bit-exact against the reference run)
    acc = np.float64(0.0)
    for t in range(1, T):
        acc = acc + np.float64(emission[t, blank_id])
        trellis[t, 0] = F32(acc)
    trellis[0, 1:] = NEG_INF
    if N > 1:
        trellis[-N + 1:, 0] = POS_INF
    else:
        # python slice [-0:] == whole column (alignment.py:394 with num_tokens == 1)
        trellis[:, 0] = POS_INF
    tok_rest = np.asarray(tokens[1:], dtype=np.int64)
    for t in range(T - 1):
        if N > 1:
            stay = trellis[t, 1:] + emission[t, blank_id]
            change = trellis[t, :-1] + token_emission(emission[t], tok_rest, blank_id)
            trellis[t + 1, 1:] = np.maximum(stay, change)
    return trellis


def backtrack_beam(trellis, emission, tokens, blank_id=0, beam_width=2):
    """alignment.py:500-579.  Returns list of (token_index, time_index, score) with
    score a python float (exp of an f32 emission), or None if the beams die out."""
    trellis = np.asarray(trellis, dtype=F32)
    emission = np.asarray(emission, dtype=F32)
    T, J = trellis.shape[0] - 1, trellis.shape[1] - 1

    def prob(x):
        return float(np.exp(F32(x)))

    # beam = (token_index, time_index, score, path)
    beams = [(J, T, float(trellis[T, J]), [(J, T, prob(emission[T, blank_id]))])]
    while beams and beams[0][0] > 0:
        nxt = []
        for (j, t, _s, path) in beams:
            if t <= 0:
                continue
            p_stay = emission[t - 1, blank_id]
            tok = tokens[j]
            p_change = wildcard_max(emission[t - 1], blank_id) if tok == -1 else emission[t - 1, tok]
            stay_score = float(trellis[t - 1, j])
            change_score = float(trellis[t - 1, j - 1]) if j > 0 else -math.inf
            if not math.isinf(stay_score):
                nxt.append((j, t - 1, stay_score, path + [(j, t - 1, prob(p_stay))]))
            if j > 0 and not math.isinf(change_score):
                nxt.append((j - 1, t - 1, change_score, path + [(j - 1, t - 1, prob(p_change))]))
        beams = sorted(nxt, key=lambda b: b[2], reverse=True)[:beam_width]
        if not beams:
            break
    if not beams:
        return None
    j, t, _s, path = beams[0]
    path = list(path)
    while t > 0:
        path.append((j, t - 1, prob(emission[t - 1, blank_id])))
        t -= 1
    return path[::-1]


def merge_repeats(path, transcript):
    """alignment.py:597-613 -> list of (label, start, end, score)."""
    i1 = i2 = 0
    segs = []
    while i1 < len(path):
        while i2 < len(path) and path[i1][0] == path[i2][0]:
            i2 += 1
        score = sum(path[k][2] for k in range(i1, i2)) / (i2 - i1)
        segs.append((transcript[path[i1][0]], path[i1][1], path[i2 - 1][1] + 1, score))
        i1 = i2
    return segs


def char_times(segs, t1, t2, n_frames):
    """alignment.py:278-290: frame index -> seconds, rounded to 3 decimals."""
    ratio = (t2 - t1) * 1 / (n_frames - 1)
    return [(round(s * ratio + t1, 3), round(e * ratio + t1, 3), round(sc, 3))
            for (_l, s, e, sc) in segs]
