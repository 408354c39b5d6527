"""CPU: host halves of the backend that need no GPU -- the DTW word bookkeeping (vectorised) against the loop it replaced,
the per-token text cache, and the contexts' fall-back when memory is short (with a stand-in for the engine)."""
import warnings

import numpy as np
import pytest

from whisperx_mlx_amd import backend as BK
from whisperx_mlx_amd.tokenizer import get_tokenizer

TOKENS_PER_SECOND = BK.TOKENS_PER_SECOND


def _loop_dtw_words(tok, text_ids, path_info):
    """the published find_alignment bookkeeping as the backend had it before it was vectorised (a plain loop)"""
    n_rows, path = path_info
    if n_rows < 2 or path.shape[1] == 0 or not text_ids:
        return []
    ti, fi = path[0], path[1]
    jumps = np.concatenate([[True], np.diff(ti) > 0])
    jump_times = fi[jumps].astype(np.float64) / TOKENS_PER_SECOND
    words, word_tokens = tok.split_to_word_tokens(text_ids)
    bounds = np.concatenate([[0], np.cumsum([len(t) for t in word_tokens])])
    res = []
    for w, a, b in zip(words, bounds[:-1], bounds[1:]):
        if a >= len(jump_times):
            break
        start = float(jump_times[a])
        end = float(jump_times[min(b, len(jump_times) - 1)])
        if w.strip():
            res.append({"word": w.strip(), "start": start, "end": max(end, start), "probability": 1.0, "tok_end": int(b)})
    return res


def test_dtw_words_vectorised_equals_the_loop():
    tok = get_tokenizer(51866)
    be = BK.WhisperHipBackend.__new__(BK.WhisperHipBackend)
    be.tokenizer = tok
    rng = np.random.default_rng(3)
    for trial in range(60):
        n_tok = int(rng.integers(1, 140))
        ids = rng.integers(300, 50000, size=n_tok).tolist()
        T = int(rng.integers(1, 1500))
        # a monotone path over (rows covered, frames): sometimes shorter than the token list (words beyond it are dropped)
        rows = int(rng.integers(1, n_tok + 2))
        ti = np.sort(rng.integers(0, rows, size=T)).astype(np.int32)
        fi = np.arange(T, dtype=np.int32)
        info = (n_tok + 1, np.stack([ti, fi]))
        assert be._dtw_words(ids, info) == _loop_dtw_words(tok, ids, info), trial
    assert be._dtw_words([], (5, np.zeros((2, 3), np.int32))) == [] and be._dtw_words([400], (1, np.zeros((2, 3), np.int32))) == []


def test_decode_token_cache_is_transparent():
    tok = get_tokenizer(51866)
    ids = [400, 401, 400, 50000, 401]
    first = [tok.decode_token(t) for t in ids]
    assert first == [tok.decode([t]) for t in ids] and [tok.decode_token(t) for t in ids] == first
    words, word_tokens = tok.split_to_word_tokens(ids)
    assert sum(len(w) for w in word_tokens) == len(ids) and "".join(words) == tok.decode(ids)


class _FakeEngine:
    """stands in for WhisperHipEngine: a context of more than `fits` rows does not get its workspace"""
    fits = 32
    made = []

    def __init__(self, dims, packed, max_batch, device_index=0, alignment_heads=None):
        if max_batch > self.fits or len(_FakeEngine.made) >= 2:
            raise RuntimeError("wx_finalize failed (rc=-1): hipMalloc failed: out of memory (api.hip:123)")
        self.max_batch, self.packed, self.alignment_heads, self.device = max_batch, packed, alignment_heads, "cpu"
        _FakeEngine.made.append(self)

    def close(self):
        pass


def test_contexts_shrink_when_memory_is_short_host_logic(monkeypatch):
    """contexts.py (the memory policy): rows are halved until a context fits, later contexts that do not fit mean fewer passes
    in flight, and errors that are not about memory are passed on"""
    import torch
    _FakeEngine.made = []
    from whisperx_mlx_amd import contexts as CX
    monkeypatch.setattr(CX, "WhisperHipEngine", _FakeEngine)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (64 << 30, 288 << 30))
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    rows = 128
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        while True:
            try:
                eng = BK._new_context(None, {}, rows, 0, None)
                break
            except RuntimeError as e:
                assert BK._is_oom(e)
                rows //= 2
        assert rows == 32 and eng.max_batch == 32
        be = BK.WhisperHipBackend.__new__(BK.WhisperHipBackend)
        be.engines, be.engine, be.dims, be.device_index = [eng], eng, None, 0
        got = be._get_engines(3)
        assert len(got) == 2 and be._no_more_contexts                       # the third context did not fit
        assert any("no memory for engine context 3" in str(x.message) for x in w)
        assert len(be._get_engines(3)) == 2                                  # and is not tried again every call
    # a context that fits but leaves torch no room counts as not fitting
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (1 << 30, 288 << 30))
    _FakeEngine.made = []
    with pytest.raises(RuntimeError, match="out of memory"):
        BK._new_context(None, {}, 16, 0, None)
    assert not BK._is_oom(RuntimeError("wx_finalize failed: weight enc.conv1.w has 12 bytes, expected 24"))


def test_extra_contexts_are_sized_by_the_job(monkeypatch):
    """contexts.get_contexts: a job of <= 16-row passes gets 16-row contexts beyond the first; the first wide job
    rebuilds exactly the contexts it uses at the first context's size, on their old streams; later small jobs reuse them"""
    import torch
    from whisperx_mlx_amd import contexts as CX

    class Eng:
        made = []

        def __init__(self, dims, packed, max_batch, device_index=0, alignment_heads=None):
            self.max_batch, self.packed, self.alignment_heads, self.device = max_batch, packed, alignment_heads, "cpu"
            self.stream, self.closed = object(), False
            Eng.made.append(self)

        def close(self):
            self.closed = True

    monkeypatch.setattr(CX, "WhisperHipEngine", Eng)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (200 << 30, 288 << 30))
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    first = CX.first_context(None, {}, 128, 0, None)
    be = BK.WhisperHipBackend.__new__(BK.WhisperHipBackend)
    be.engines, be.engine, be.dims, be.device_index = [first], first, None, 0
    small = be._get_engines(4, rows=16)
    assert [e.max_batch for e in small] == [128, 16, 16, 16]
    streams = [e.stream for e in small]
    wide = be._get_engines(3, rows=112)
    assert [e.max_batch for e in wide] == [128, 128, 128] and [e.stream for e in wide] == streams[:3]
    assert small[1].closed and small[2].closed and not small[3].closed and be.engines[3] is small[3]
    again = be._get_engines(4, rows=16)
    assert [e.max_batch for e in again] == [128, 128, 128, 16] and len(Eng.made) == 6       # nothing new for the second small job
    assert [e.max_batch for e in be._get_engines(2)] == [128, 128]                          # rows=None: full-size contexts
