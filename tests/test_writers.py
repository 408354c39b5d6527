"""Writers: byte-identical to the reference's gold outputs given the gold result dict.

Fixtures are DATA files of the reference (`whisperx-large-v3-gold-standard/30m.{json,srt,vtt,tsv,txt}`,
copied to tests/golden/gold30m/, the JSON gzipped): the JSON is the result dict the reference's own
writers turned into the other four files (default options: no line limits, no highlighting)."""
import gzip
import io
import json
import os

import pytest

from whisperx_mlx_amd import writers as W

GOLD = os.path.join(os.path.dirname(__file__), "golden", "gold30m")
DEFAULTS = {"max_line_width": None, "max_line_count": None, "highlight_words": False}


@pytest.fixture(scope="module")
def gold():
    with gzip.open(os.path.join(GOLD, "30m.json.gz"), "rt", encoding="utf-8") as f:
        return json.load(f)


@pytest.mark.parametrize("ext", ["txt", "vtt", "srt", "tsv"])
def test_writer_bytes_equal_gold(gold, ext, tmp_path):
    W.get_writer(ext, str(tmp_path))(gold, "/some/dir/30m.wav", DEFAULTS)
    got = open(tmp_path / f"30m.{ext}", encoding="utf-8").read()
    ref = open(os.path.join(GOLD, f"30m.{ext}"), encoding="utf-8").read()
    assert got == ref


def test_all_and_json_roundtrip(gold, tmp_path):
    W.get_writer("all", str(tmp_path))(gold, "30m.mp3", DEFAULTS)
    assert sorted(os.listdir(tmp_path)) == ["30m.json", "30m.srt", "30m.tsv", "30m.txt", "30m.vtt"]
    assert json.load(open(tmp_path / "30m.json", encoding="utf-8")) == gold


def test_format_timestamp():
    assert W.format_timestamp(0) == "00:00.000"
    assert W.format_timestamp(3661.001, always_include_hours=True, decimal_marker=",") == "01:01:01,001"
    assert W.format_timestamp(3661.001) == "01:01:01.001"          # hours appear once they are non-zero
    assert W.format_timestamp(59.9996) == "01:00.000"
    with pytest.raises(AssertionError):
        W.format_timestamp(-1)


def _cues(result, **opt):
    w = W.WriteVTT(".")
    return list(w.iterate_result(result, {**DEFAULTS, **opt}))


def test_line_limits_pauses_and_highlight():
    words = [{"word": w, "start": s, "end": s + 0.3} for w, s in
             (("alpha", 0.0), ("beta", 0.5), ("gamma", 1.0), ("delta", 6.0), ("epsilon", 6.5))]
    res = {"language": "en", "segments": [{"start": 0.0, "end": 7.0, "text": "alpha beta gamma delta epsilon", "words": words}]}
    # no limits: one cue per segment
    assert _cues(res) == [("00:00.000", "00:07.000", "alpha beta gamma delta epsilon")]
    # width 10, 2 lines per cue; the 5 s pause before "delta" also breaks the cue
    got = _cues(res, max_line_width=10, max_line_count=2)
    # (words are joined by " ", a line break is a "\n" glued to the front of the next word -- as the reference does)
    assert [t for _, _, t in got] == ["alpha beta \ngamma", "delta \nepsilon"]
    # highlight: one cue per word, the word underlined, gaps show the plain line
    hl = _cues(res, highlight_words=True)
    assert hl[0] == ("00:00.000", "00:00.300", "<u>alpha</u> beta gamma delta epsilon")
    assert hl[1] == ("00:00.300", "00:00.500", "alpha beta gamma delta epsilon")
    assert sum("<u>" in t for _, _, t in hl) == 5
    # words without timing (numerals the aligner could not place) stay in the text
    res2 = {"language": "en", "segments": [{"start": 1.0, "end": 2.0, "text": "in 2024", "speaker": "SPEAKER_00",
                                            "words": [{"word": "in", "start": 1.0, "end": 1.2}, {"word": "2024"}]}]}
    assert _cues(res2) == [("00:01.000", "00:02.000", "[SPEAKER_00]: in 2024")]
    # languages written without spaces join the words directly
    res3 = {"language": "ja", "segments": [{"start": 0.0, "end": 1.0, "text": "こんにちは",
                                            "words": [{"word": "こん", "start": 0.0, "end": 0.4}, {"word": "にちは", "start": 0.4, "end": 1.0}]}]}
    assert _cues(res3)[0][2] == "こんにちは"


def test_unaligned_segments_tsv_aud(tmp_path):
    res = {"language": "en", "segments": [{"start": 0.0, "end": 1.5, "text": " a --> b\tc ", "speaker": "S1"}]}
    assert _cues(res) == [("00:00.000", "00:01.500", "[S1]: a -> b\tc")]
    buf = io.StringIO()
    W.WriteTSV(".").write_result(res, buf, {})
    assert buf.getvalue() == "start\tend\ttext\n0\t1500\ta --> b c\n"
    buf = io.StringIO()
    W.WriteAudacity(".").write_result(res, buf, {})
    assert buf.getvalue() == "0.0\t1.5\t[[S1]]a --> b c\n"
    buf = io.StringIO()
    W.WriteTXT(".").write_result(res, buf, {})
    assert buf.getvalue() == "[S1]: a --> b\tc\n"
    with pytest.raises(ValueError):
        W.get_writer("docx", ".")


def test_option_sets_equal_reference_outputs():
    """tests/golden/writers_opts.json: outputs of the reference's own writers (tools/make_golden.py::make_writers)
    for line limits, highlighting, a speaker label, a word without timing and a language without spaces."""
    doc = json.load(open(os.path.join(os.path.dirname(GOLD), "writers_opts.json"), encoding="utf-8"))
    classes = {"srt": W.WriteSRT, "vtt": W.WriteVTT, "tsv": W.WriteTSV, "txt": W.WriteTXT, "aud": W.WriteAudacity}
    assert len(doc["cases"]) == 10
    for case in doc["cases"]:
        result = doc["results"][case["result"]]
        for ext, ref in case["out"].items():
            buf = io.StringIO()
            classes[ext](".").write_result(result, buf, case["options"])
            assert buf.getvalue() == ref, (case["result"], case["options"], ext)
