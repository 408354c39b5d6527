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
