"""CLI task flow on the CPU: flags, language handling, the transcribe -> align -> writers
sequence with the GPU stages replaced by canned results (the stages themselves are covered by the GPU tests)."""
import json
import os
import wave

import numpy as np
import pytest

import whisperx_mlx_amd.alignment as A
import whisperx_mlx_amd.backend as B
from whisperx_mlx_amd import transcribe as T


def _wav(path, seconds=1.0, rate=16000, channels=1):
    t = np.arange(int(seconds * rate)) / rate
    x = (0.3 * np.sin(2 * np.pi * 440 * t) * 32767).astype(np.int16)
    if channels > 1:
        x = np.repeat(x[:, None], channels, axis=1).reshape(-1)
    with wave.open(str(path), "wb") as w:
        w.setnchannels(channels)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes(x.tobytes())


def test_load_audio_without_ffmpeg(tmp_path, monkeypatch):
    monkeypatch.setenv("PATH", str(tmp_path))          # no ffmpeg binary
    _wav(tmp_path / "a.wav", 0.5)
    a = B.load_audio(str(tmp_path / "a.wav"))
    assert a.dtype == np.float32 and a.shape == (8000,) and 0.29 < np.abs(a).max() < 0.31
    _wav(tmp_path / "b.wav", 0.5, rate=48000, channels=2)
    b = B.load_audio(str(tmp_path / "b.wav"))
    assert b.shape == (8000,) and 0.25 < np.abs(b).max() < 0.35
    np.save(tmp_path / "c.npy", a)
    assert np.array_equal(B.load_audio(str(tmp_path / "c.npy")), a)


def test_flags_and_language():
    p = T.build_parser()
    a = p.parse_args(["x.wav", "--model", "large-v3", "--language", "German", "--batch_size", "16", "-f", "vtt",
                      "--max_line_width", "42", "--max_line_count", "2", "--highlight_words", "True", "--no_align"]).__dict__
    assert a["audio"] == ["x.wav"] and a["output_format"] == "vtt" and a["no_align"] and a["highlight_words"] is True
    assert T._normalise_language("German") == "de" and T._normalise_language("yue") == "yue" and T._normalise_language(None) is None
    with pytest.raises(ValueError):
        T._normalise_language("klingon")
    with pytest.raises(SystemExit):
        p.parse_args(["x.wav", "--output_format", "docx"])


class _FakePipeline:
    def __init__(self, log):
        self.log = log

    def transcribe(self, audio, **kw):
        self.log.append(("transcribe", len(audio), kw.get("batch_size"), kw.get("language")))
        return {"segments": [{"start": 0.0, "end": 1.0, "text": " hello world"}], "language": "en"}


def test_task_flow_with_alignment(tmp_path, monkeypatch):
    log = []
    monkeypatch.setenv("PATH", str(tmp_path))
    _wav(tmp_path / "clip.wav", 1.0)
    monkeypatch.setattr(B, "load_model", lambda name, **kw: (log.append(("load_model", name, kw["batch_size"], kw["backend"])), _FakePipeline(log))[1])
    monkeypatch.setattr(A, "load_align_model", lambda lang, device, model_name=None, model_dir=None: (log.append(("load_align", lang)), ("W2V", {"language": lang}))[1])

    def fake_align(segments, model, meta, audio, device, **kw):
        log.append(("align", len(segments), kw["interpolate_method"]))
        words = [{"word": "hello", "start": 0.1, "end": 0.4, "score": 0.9}, {"word": "world", "start": 0.5, "end": 0.9, "score": 0.8}]
        return {"segments": [{"start": 0.1, "end": 0.9, "text": "hello world", "words": words}], "word_segments": words}
    monkeypatch.setattr(A, "align", fake_align)
    out = tmp_path / "out"
    T.cli([str(tmp_path / "clip.wav"), "--model", "tiny", "--output_dir", str(out), "-f", "all", "--batch_size", "4", "--verbose", "False"])
    assert [e[0] for e in log] == ["load_model", "transcribe", "load_align", "align"]
    assert log[0][1:] == ("tiny", 4, "hip") and log[1][1] == 16000 and log[3][2] == "nearest"
    assert sorted(os.listdir(out)) == ["clip.json", "clip.srt", "clip.tsv", "clip.txt", "clip.vtt"]
    assert open(out / "clip.srt").read() == "1\n00:00:00,100 --> 00:00:00,900\nhello world\n\n"
    assert json.load(open(out / "clip.json"))["language"] == "en"


def test_task_flow_no_align_and_rejections(tmp_path, monkeypatch):
    log = []
    monkeypatch.setenv("PATH", str(tmp_path))
    _wav(tmp_path / "clip.wav", 0.2)
    monkeypatch.setattr(B, "load_model", lambda name, **kw: _FakePipeline(log))
    T.cli([str(tmp_path / "clip.wav"), "--no_align", "--output_dir", str(tmp_path), "-f", "tsv", "--verbose", "False"])
    assert open(tmp_path / "clip.tsv").read() == "start\tend\ttext\n0\t1000\thello world\n"
    with pytest.raises(SystemExit):      # word-level options need word timings
        T.cli([str(tmp_path / "clip.wav"), "--no_align", "--highlight_words", "True", "--output_dir", str(tmp_path)])
    with pytest.raises(SystemExit):
        T.cli([str(tmp_path / "clip.wav"), "--diarize", "--output_dir", str(tmp_path)])


def test_suppress_tokens_follow_the_vocabulary():
    """SuppressTokens("-1"): the multilingual id list must not be applied to the English-only (gpt2) vocabulary, and a
    checkpoint's own tokenizer.json decides when there is one (published non_speech_tokens construction)."""
    from whisperx_mlx_amd import tokenizer as TK
    multi, en = TK.get_tokenizer(51865), TK.get_tokenizer(51864)
    assert not en.is_multilingual and en.eot == 50256 and en.sot_sequence() == [50257]
    assert (en.translate, en.transcribe, en.no_speech, en.no_timestamps, en.timestamp_begin) == (50357, 50358, 50361, 50362, 50363)
    assert (multi.translate, multi.transcribe, multi.no_speech, multi.no_timestamps, multi.timestamp_begin) == (50358, 50359, 50362, 50363, 50364)
    assert en.timestamp_begin + 1501 == 51864 and multi.timestamp_begin + 1501 == 51865      # the layouts fill their vocabularies
    sm, se = set(multi.suppress_tokens()), set(en.suppress_tokens())
    assert 359 in sm and 359 not in se and 357 in se and 357 not in sm           # "[" family: different ids per vocabulary
    assert {en.sot, en.sot_prev, en.sot_lm, en.no_speech, en.transcribe, en.translate} <= se
    # derivation from a vocabulary: single-token symbols (bare or with a leading space), note symbols by first token
    vocab = {" -": 5, " '": 6, "(": 7, " (": 8, "((": 9, "♪": 10, "x": 11, " ": 12, "[": 13}

    def encode(text):      # greedy longest match over the toy vocabulary
        out, i = [], 0
        while i < len(text):
            for L in (3, 2, 1):
                if text[i: i + L] in vocab:
                    out.append(vocab[text[i: i + L]])
                    i += L
                    break
            else:
                out.append(99)
                i += 1
        return out
    got = TK.non_speech_tokens_from(encode)
    assert {5, 6, 7, 8, 9, 10, 13} <= set(got) and 11 not in got
    assert 12 in got            # " ♪" -> [" ", note]: a note symbol is banned by its FIRST token, as published
