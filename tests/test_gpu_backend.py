"""-m gpu: the drop-in surface (WhisperBackend API / load_model / pipeline) on the GPU with a
random-init tiny model: result-dict contract of whisperx/types.py:4-69 and the call shapes
of whisperx/asr.py:28-120.  (The reference's own tests assert exactly these keys:
tests/test_mlx_backend.py:24-307.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.synth import speechlike_audio          # noqa: E402
from whisperx_mlx_amd import backend as BK        # noqa: E402


def _pipe():
    if not hasattr(_pipe, "p"):
        _pipe.p = BK.load_model("tiny", device="cuda", backend="hip", batch_size=8, random_init=True, seed=1)
    return _pipe.p


def test_transcribe_result_contract():
    p = _pipe()
    audio = speechlike_audio(70.0, seed=3)
    res = p.transcribe(audio, batch_size=8, language="en")
    assert set(res) >= {"segments", "language"} and res["language"] == "en"
    assert 1 <= len(res["segments"]) <= 3
    last_end = 0.0
    for s in res["segments"]:
        assert {"start", "end", "text", "id"} <= set(s)
        assert 0.0 <= s["start"] < s["end"] <= 70.0 + 1e-6 and s["start"] >= last_end - 1e-6
        last_end = s["end"]
        assert isinstance(s["text"], str) and s["text"]


def test_transcribe_batch_vad_segments_and_dtw_words():
    p = _pipe()
    audio = speechlike_audio(50.0, seed=4)
    vad = BK.merge_chunks([(0.5, 9.0), (9.5, 21.0), (22.0, 31.5), (33.0, 49.0)], 30)
    assert [(round(v["start"], 1), round(v["end"], 1)) for v in vad] == [(0.5, 21.0), (22.0, 49.0)]
    segs = [dict(v, audio=audio[int(v["start"] * 16000): int(v["end"] * 16000)]) for v in vad]
    res = p.backend.transcribe_batch(segs, batch_size=8, language="en", word_timestamps="dtw")
    assert len(res["segments"]) == 2
    for s, v in zip(res["segments"], vad):
        assert v["start"] - 1e-6 <= s["start"] and s["end"] <= v["end"] + 1e-6
        assert "words" in s
        prev = s["start"]
        for w in s["words"]:
            assert {"word", "start", "end", "probability"} <= set(w)
            assert prev - 1e-6 <= w["start"] <= w["end"] <= s["end"] + 1e-6
            prev = w["start"]


def test_detect_language_and_properties():
    p = _pipe()
    lang = p.detect_language(speechlike_audio(5.0, seed=5))
    assert lang in p.backend.supported_languages and len(p.backend.supported_languages) == 99
    assert p.backend.is_multilingual


def test_same_audio_same_tokens_across_batch_positions():
    """a chunk's result must not depend on where it sits in the batch (padding/ragged lengths)"""
    be = _pipe().backend
    a = speechlike_audio(12.0, seed=6)
    b = speechlike_audio(30.0, seed=7)
    r1 = be._decode_chunks([a, b, a], "en", "transcribe", False)
    r2 = be._decode_chunks([a], "en", "transcribe", False)
    assert r1[0]["tokens"] == r1[2]["tokens"] == r2[0]["tokens"]


def test_cli_end_to_end_random_weights(tmp_path, monkeypatch):
    """`python -m whisperx_mlx_amd clip.wav ...` (transcribe.py): flags -> load_model -> transcribe -> writers on the
    GPU, ffmpeg-less wav input, seeded random tiny weights (no checkpoint ships); the files must agree with each other."""
    import json
    import wave
    from whisperx_mlx_amd import transcribe as T
    monkeypatch.setenv("PATH", str(tmp_path))          # take the ffmpeg-less path whatever the box has
    x = (speechlike_audio(35.0, seed=5) * 32767).astype(np.int16)
    with wave.open(str(tmp_path / "clip.wav"), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(x.tobytes())
    out = tmp_path / "out"
    T.cli([str(tmp_path / "clip.wav"), "--model", "tiny", "--random_init", "True", "--word_timestamps", "True",
           "--output_dir", str(out), "-f", "all", "--batch_size", "8", "--language", "en", "--verbose", "False"])
    res = json.load(open(out / "clip.json"))
    assert res["language"] == "en" and len(res["segments"]) >= 1
    tsv = open(out / "clip.tsv").read().splitlines()
    assert tsv[0] == "start\tend\ttext" and len(tsv) == 1 + len(res["segments"])
    assert open(out / "clip.txt").read().count("\n") == len(res["segments"])
    assert open(out / "clip.vtt").read().startswith("WEBVTT\n\n")
    for seg in res["segments"]:
        assert 0.0 <= seg["start"] <= seg["end"] <= 35.5
    # --word_timestamps True = word times from the decoder's cross-attention (DTW): words must be there
    assert all(seg.get("words") for seg in res["segments"])
    for seg in res["segments"]:
        for w in seg["words"]:
            assert seg["start"] - 1e-6 <= w["start"] <= w["end"] <= seg["end"] + 1e-6


def test_edge_inputs():
    """empty / very short / exactly one window / several windows / no segments / one-row batches / the longest
    allowed decode / a decode that would overrun n_text_ctx."""
    from whisperx_mlx_amd._lib import WxError
    from whisperx_mlx_amd.tokenizer import get_tokenizer
    p = _pipe()
    for audio, n in ((np.zeros(0, np.float32), 1), (speechlike_audio(0.05, seed=1), 1), (speechlike_audio(30.0, seed=3), 1),
                     (speechlike_audio(61.0, seed=4), 3)):
        res = p.transcribe(audio, batch_size=2, language="en")
        assert len(res["segments"]) == n and res["language"] == "en"
        for seg in res["segments"]:
            assert 0.0 <= seg["start"] <= seg["end"] and isinstance(seg["text"], str)
    assert p.transcribe(speechlike_audio(12.3, seed=2), batch_size=1, language="en")["segments"][0]["end"] <= 12.31
    assert p.backend.transcribe_batch([], batch_size=8, language="en")["segments"] == []
    eng = p.backend.engine
    tok = get_tokenizer(eng.dims.n_vocab)
    enc = eng.encode((torch.randn(2, 3000, eng.dims.n_mels) * 0.5).half().cuda())
    out = eng.decode(enc, tok, tok.sot_sequence(), rules=0, forced_len=eng.dims.n_text_ctx // 2, capture_qk=True)
    assert out.n_sampled == eng.dims.n_text_ctx // 2          # 224: every capture row of the DTW score buffer used
    paths = eng.dtw_path(out, tok.eot)
    assert len(paths) == 2
    with pytest.raises(WxError):
        eng.decode(enc, tok, tok.sot_sequence(), rules=0, forced_len=eng.dims.n_text_ctx - 2)
    eng.check_status()


# ------------------------------------------------------------------------------- scheduler + DTW word assembly
def _chunks(n, seed0=40):
    lens = [30.0, 12.5, 30.0, 7.0, 21.3, 30.0, 3.2, 30.0]
    return [speechlike_audio(lens[i % len(lens)], seed=seed0 + i) for i in range(n)]


def test_scheduler_tokens_equal_single_engine():
    """the product's scheduler (3 engine contexts + launcher threads, double-buffered result slots, passes of different
    sizes: one context cuts 37 chunks into 8/8/7/7/7 rows, three contexts into 8/7/8/7/7) returns exactly what one
    engine returns pass after pass: tokens, log-probabilities and DTW words"""
    be = _pipe().backend
    chunks = _chunks(37)                                   # 8 rows per pass at most: 5 passes (backend.pass_sizes)
    one = be._decode_chunks(chunks, "en", "transcribe", "dtw", passes_in_flight=1)
    assert len(be.engines) >= 1
    many = be._decode_chunks(chunks, "en", "transcribe", "dtw", passes_in_flight=3)
    assert len(be.engines) >= 3 and len(one) == len(many) == 37
    for a, b in zip(one, many):
        assert a["tokens"] == b["tokens"] and a["sum_logprob"] == b["sum_logprob"] and a["words"] == b["words"]
    again = be._decode_chunks(chunks, "en", "transcribe", "dtw")        # graphs warm: no serial head passes
    assert [r["tokens"] for r in again] == [r["tokens"] for r in one]
    # chunks already resident in HBM (what bench.py hands over) decode to the same tokens as host arrays
    dev = [torch.from_numpy(c).cuda() for c in chunks[:11]]
    on_dev = be._decode_chunks(dev, "en", "transcribe", False)
    assert [r["tokens"] for r in on_dev] == [r["tokens"] for r in one[:11]]
    res = be.transcribe_batch([{"start": 0.0, "end": len(c) / 16000.0, "audio": c} for c in chunks[:9]], language="en",
                              word_timestamps="dtw", return_chunks=True)
    assert [c["tokens"] for c in res["chunks"]] == [r["tokens"] for r in one[:9]]
    # the scheduler decodes longest chunks first; results come back in the caller's order whatever the lengths
    mixed = [chunks[i] for i in (3, 0, 6, 1, 4)]                       # 7 s, 30 s, 3.2 s, 12.5 s, 21.3 s
    got = be._decode_chunks(mixed, "en", "transcribe", False, rows_per_pass=2, passes_in_flight=2)
    assert [r["tokens"] for r in got] == [one[i]["tokens"] for i in (3, 0, 6, 1, 4)]


@pytest.mark.parametrize("variant", ["upstream", "inrepo"])
def test_dtw_words_against_oracle(variant, monkeypatch):
    """backend._dtw_words / _dtw_words_inrepo (host word assembly over the device DTW path) against the oracle's
    restatement on the oracle's own alignment matrix and DP: every word within +-20 ms (north_star tolerance)."""
    from oracle import dtw as ODTW
    be = _pipe().backend
    eng, tok = be.engine, be.tokenizer
    # words of one to three tokens: a token id divisible by 3 opens a word (the placeholder vocabulary has no spaces)
    monkeypatch.setattr(tok, "decode_token", lambda t: (" " if t % 3 == 0 else "") + f"t{t}", raising=False)
    chunks = _chunks(6, seed0=70)
    n = len(chunks)
    pcm = torch.zeros(n, 480000)
    for i, c in enumerate(chunks):
        pcm[i, : len(c)] = torch.from_numpy(c)
    nv = torch.tensor([len(c) for c in chunks], dtype=torch.int32)
    enc = eng.encode(eng.logmel(pcm.cuda(), nv.cuda()))
    # a seeded random model under the default rules emits next to no text: only 60 text ids and EOT may be sampled here
    # (SuppressTokens on everything else), so rows are 1 .. 224 text tokens long and most of them end with EOT
    from whisperx_mlx_amd import engine as E
    allowed = set(range(3000, 3060)) | {tok.eot}
    dec = eng.decode(enc, tok, tok.sot_sequence("en", "transcribe"), rules=E.RULE_SUPPRESS_TOKENS,
                     suppress_ids=[t for t in range(eng.dims.n_vocab) if t not in allowed], capture_qk=True)
    nfr = torch.clamp((nv + 319) // 320, min=8, max=1500)
    qk = eng.align_qk(n).cpu().numpy()
    paths = eng.dtw_path(dec, tok.eot, mode=1 if variant == "inrepo" else 0, n_frames=nfr)
    eng.check_status()
    toks = dec.tokens.cpu().numpy()
    P, S = dec.n_prompt, dec.n_sampled
    n_words = 0
    for b in range(n):
        sampled = toks[b, P: P + S].tolist()
        has_eot = tok.eot in sampled
        seq = sampled[: sampled.index(tok.eot)] if has_eot else sampled
        text_ids = [t for t in seq if t < tok.eot]
        rows = [s for s, t in enumerate(seq) if t < tok.eot] + ([len(seq)] if has_eot else [])
        if len(text_ids) < 2:
            continue
        sel = qk[b][:, rows, : int(nfr[b])]
        words, word_tokens = tok.split_to_word_tokens(text_ids)
        if variant == "upstream":
            got = be._dtw_words(text_ids, paths[b])
            # the matrix carries one extra trailing row for EOT; a row that ran to sample_len has none, so its last
            # word's end boundary is not defined by the published bookkeeping and is left out of the comparison
            counts = [len(t) for t in word_tokens]
            st, en = ODTW.word_times_upstream(ODTW.alignment_matrix_upstream(sel), counts if has_eot else counts[:-1])
            ref = [(w.strip(), s, max(e, s)) for w, s, e in zip(words, st, en)]
            got = got[: len(ref)]
        else:
            got = be._dtw_words_inrepo(text_ids, paths[b])
            ref = ODTW.word_times_inrepo(ODTW.inrepo_row0(ODTW.alignment_matrix_inrepo(sel)), [tok.decode_token(t) for t in text_ids])
        assert [g["word"] for g in got] == [r[0] for r in ref], b
        for g, r in zip(got, ref):
            assert abs(g["start"] - r[1]) <= 0.020 + 1e-9 and abs(g["end"] - r[2]) <= 0.020 + 1e-9, (b, g, r)
            n_words += 1
    assert n_words >= 60


def test_longform_batch_processor_int8_on_the_gpu():
    """config 5's path at test size: a 100 s stream through batch_processor.batch_transcribe (30 s chunks with 0.5 s
    overlap, reference whisperx/batch_processor.py:279-338) on a backend with int8 decoder GEMV weights -- the chunker
    hands the whole chunk list to the scheduler; the per-chunk tokens must equal a direct decode of the same chunks."""
    from whisperx_mlx_amd.batch_processor import BatchProcessor, batch_transcribe
    be = BK.WhisperHipBackend("tiny", random_init=True, seed=3, compute_type="int8", max_batch=4)
    assert any(k.endswith(".wq") for k in be.engine.packed) and be.compute_type == "int8"
    audio = speechlike_audio(100.0, seed=9)
    segs = [{"start": 0.0, "end": 100.0}]
    out = batch_transcribe(audio, segs, be, batch_size=4, decode_options={"language": "en"})
    assert len(out) == 1 and out[0]["start"] == 0.0 and out[0]["end"] == 100.0 and isinstance(out[0]["text"], str)
    chunks = BatchProcessor(batch_size=4).create_chunks(audio, segs)
    assert [round(c.start_time, 1) for c in chunks] == [0.0, 29.5, 59.0, 88.5] and len(chunks[-1].audio) == int(11.5 * 16000)
    direct = be._decode_chunks([c.audio for c in chunks], "en", "transcribe", False, passes_in_flight=1)
    again = be._decode_chunks([c.audio for c in chunks], "en", "transcribe", False, rows_per_pass=2, passes_in_flight=2)
    assert [r["tokens"] for r in direct] == [r["tokens"] for r in again]
    merged = BatchProcessor()._merge_overlapping_text(list(zip(chunks, [{"text": r["text"]} for r in direct])))
    assert out[0]["text"] == merged


def test_concurrent_callers_take_turns():
    """two user threads calling the same backend: scheduler runs are serialised over the shared engine contexts (a
    context is single-threaded), both get the single-caller result"""
    import threading
    be = _pipe().backend
    chunks = _chunks(10, seed0=90)
    ref = [r["tokens"] for r in be._decode_chunks(chunks, "en", "transcribe", False)]
    out, errs = {}, []

    def call(k):
        try:
            out[k] = [r["tokens"] for r in be._decode_chunks(chunks if k == 0 else chunks[::-1], "en", "transcribe", False)]
        except BaseException as e:     # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=call, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert out[0] == ref and out[1] == ref[::-1]


def test_default_passes_in_flight_are_checked_against_the_hardware_queues():
    """Four passes in flight need four streams that run side by side.  tests/conftest.py asks for 8 hardware queues before
    the GPU is touched, so here the backend's default must settle on 4 and the engine streams must overlap; a stream
    list that repeats ONE stream (what two streams on one hardware queue amount to) must read as serialised."""
    import ctypes as C
    from whisperx_mlx_amd import HW_QUEUES, _lib
    be = _pipe().backend
    L = _lib.lib()

    def factor(handles):
        arr = (C.c_void_p * len(handles))(*handles)
        f = C.c_float(0.0)
        assert L.wx_streams_overlap(0, arr, len(handles), 300, C.byref(f)) == 0
        return f.value

    n = be._default_lanes(8)                # asks the streams; swaps in streams that do run side by side
    assert factor([e._s for e in be._get_engines(n)]) < 1.5
    if HW_QUEUES >= 6 and not be._lanes_req:
        assert n == 4 and be._default_lanes(48) == 3
    e0, e1 = be._get_engines(2)
    assert factor([e0._s, e1._s]) < 1.5
    assert factor([e0._s, e0._s]) > 1.7                      # the same stream twice: one after the other
    assert L.wx_streams_overlap(0, None, 2, 300, C.byref(C.c_float())) != 0


def test_a_chunk_decodes_the_same_in_every_job():
    """jobs of 1 .. 45 chunks through the default scheduler (passes in flight settled by the backend, every pass launched
    with batch_size rows, the remainder as padding rows): each chunk gets the tokens, log-probability and DTW words it
    gets when the 45 chunks are decoded pass after pass on one context"""
    be = _pipe().backend
    chunks = _chunks(45, seed0=300)
    ref = be._decode_chunks(chunks, "en", "transcribe", "dtw", passes_in_flight=1)
    for n in (1, 7, 9, 17, 33, 45):
        idx = [(7 * i + n) % 45 for i in range(n)]
        out = be._decode_chunks([chunks[j] for j in idx], "en", "transcribe", "dtw")
        for o, j in zip(out, idx):
            assert o["tokens"] == ref[j]["tokens"] and o["sum_logprob"] == ref[j]["sum_logprob"] and o["words"] == ref[j]["words"], (n, j)


def test_give_up_recovery_through_the_scheduler(monkeypatch):
    """the recovery of `_decode_chunks` from a key-split give-up, through the product path.  In the middle
    of a job (three passes in flight, the second pass of a context) the device flag of that context is raised on its
    stream exactly as a decode kernel whose bounded wait expired raises it (wx_test_raise_device_flag).  The scheduler must
    notice at the end of the lane's run, warn, decode the WHOLE job again without key splits -- once -- and return that
    decode's results: equal, chunk by chunk, to a job decoded without key splits from the start.  The flag is cleared
    (the next job runs with the default two splits again and matches the default result); after three such events the
    backend stays without key splits."""
    import ctypes as C
    import warnings
    from whisperx_mlx_amd import _lib
    be = _pipe().backend
    chunks = _chunks(60, seed0=500)
    kw = dict(rows_per_pass=8, passes_in_flight=3, forced_len=12)
    default = be._decode_chunks(chunks, "en", "transcribe", "dtw", **kw)
    nosplit = be._decode_chunks(chunks, "en", "transcribe", "dtw", _force_split=1, **kw)
    assert be.split_giveups == 0
    real = BK.WhisperHipBackend._enqueue_pass
    calls = {"n": 0, "raised": 0}

    def enqueue_and_fail_once(self, eng, slot, batch, *a, **k):
        real(self, eng, slot, batch, *a, **k)
        calls["n"] += 1
        if calls["n"] == 5 and not calls["raised"]:          # the fifth pass enqueued of the first attempt: mid-job
            calls["raised"] = 1
            _lib.check(eng.ctx, _lib.lib().wx_test_raise_device_flag(eng.ctx, C.c_void_p(eng.stream.cuda_stream)), "raise")

    monkeypatch.setattr(BK.WhisperHipBackend, "_enqueue_pass", enqueue_and_fail_once)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = be._decode_chunks(chunks, "en", "transcribe", "dtw", **kw)
    monkeypatch.setattr(BK.WhisperHipBackend, "_enqueue_pass", real)
    assert calls["raised"] == 1 and be.split_giveups == 1
    assert len([x for x in w if "gave up" in str(x.message)]) == 1                     # decoded again exactly once
    n_pass = len(BK.pass_sizes(60, 8, 3))
    assert calls["n"] == 2 * n_pass                                                    # the whole job twice, nothing more
    assert [r["tokens"] for r in out] == [r["tokens"] for r in nosplit]
    assert [r["sum_logprob"] for r in out] == [r["sum_logprob"] for r in nosplit]
    assert [r["words"] for r in out] == [r["words"] for r in nosplit]
    for e in be.engines:
        e.check_status()                                                               # the flag was read and cleared
    again = be._decode_chunks(chunks, "en", "transcribe", "dtw", **kw)
    assert [r["tokens"] for r in again] == [r["tokens"] for r in default] and be.cross_split == 0
    be.split_giveups = 0


def test_contexts_shrink_when_memory_is_short():
    """a 128-row engine context of large-v3 holds 49 GB of workspace; on a GPU that does not have it (other processes,
    other models) the backend builds smaller contexts, and fewer of them, instead of failing -- and decodes the same tokens"""
    import gc
    import warnings
    gc.collect()
    torch.cuda.empty_cache()
    free, _total = torch.cuda.mem_get_info()
    hold = torch.empty(max(0, free - (40 << 30)), dtype=torch.uint8, device="cuda")      # leave ~40 GB
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            be = BK.WhisperHipBackend("large-v3", random_init=True, seed=4242, max_batch=16)
            assert be.engine.max_batch in (16, 32, 64) and be.rows_per_pass == be.engine.max_batch     # 128 rows did not fit
            from whisperx_mlx_amd.synth import synth_audio
            chunks = [synth_audio(300 + i, 480000 - 16000 * (i % 5)) for i in range(52)]
            kw = dict(forced_len=10)
            out = be._decode_chunks(chunks, "en", "transcribe", False, **kw)
            assert 1 <= len(be.engines) <= 4 and be.last_plan["passes_in_flight"] == min(len(be.engines), len(be.last_plan["rows"]))
            assert any("no memory for an engine context of 128 rows" in str(x.message) for x in w)
            if len(be.engines) < 3:              # no room for every context the plan wanted: fewer passes in flight, no error
                assert any("no memory for engine context" in str(x.message) for x in w)
        direct = be._decode_chunks(chunks, "en", "transcribe", False, rows_per_pass=16, passes_in_flight=1, **kw)
        assert [r["tokens"] for r in out] == [r["tokens"] for r in direct]
        be.engine.check_status()
    finally:
        del hold
        for e in list(getattr(locals().get("be"), "engines", [])):
            e.close()
        torch.cuda.empty_cache()
