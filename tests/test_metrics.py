"""CPU: the headline accuracy metric (whisperx_mlx_amd/metrics.py) on the gold-standard transcript the reference ships
(whisperx-large-v3-gold-standard/30m.json, committed as tests/golden/gold30m/30m.json.gz): identity, a known shift, words
dropped / inserted, punctuation and case."""
import copy
import gzip
import json
import os

from whisperx_mlx_amd import metrics as M

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gold30m", "30m.json.gz")


def _gold():
    with gzip.open(GOLD, "rt") as f:
        return json.load(f)


def test_gold_against_itself_and_shifted():
    g = _gold()
    ref = M.flatten_words(g)
    assert len(g["segments"]) == 779 and len(ref) == 5510
    r = M.word_mae_ms(ref, ref)
    assert r["mae_ms"] == 0.0 and r["matched"] == 1.0 and r["within_20ms"] == 1.0
    shifted = [dict(w, start=w["start"] + 0.030, end=w["end"] + 0.010) for w in ref]
    r = M.word_mae_ms(shifted, ref)
    assert r["start_mae_ms"] == 30.0 and r["end_mae_ms"] == 10.0 and r["mae_ms"] == 20.0 and r["within_20ms"] == 0.0


def test_matching_survives_edits():
    ref = M.flatten_words(_gold())[:400]
    got = copy.deepcopy(ref)
    del got[10:13]                                            # three words missing
    got.insert(50, {"word": "uh", "start": 0.0, "end": 0.1})  # one inserted
    got[100]["word"] = got[100]["word"].upper() + "!"         # case / punctuation do not break a match
    for w in got[200:]:
        w["start"] += 0.004
    r = M.word_mae_ms(got, ref)
    assert 0.985 <= r["matched"] < 1.0 and r["within_20ms"] >= 0.99    # a repeated word may pair across an edit
    assert 0.0 < r["start_mae_ms"] < 8.0 and r["end_mae_ms"] < 5.0     # one mispaired repeated word in 400
    assert M.word_mae_ms([], ref)["mae_ms"] is None
    assert M.token_similarity([1, 2, 3, 4], [1, 2, 3, 4]) == 1.0 and M.token_similarity([1, 2, 3, 4], [1, 2, 9, 4]) == 0.75
