"""Accuracy metrics of the BASELINE.json headline: word-timestamp MAE against a gold transcript, and token-list
similarity.  Definitions follow the reference's own comparison script: words are matched in order on their normalised
text with difflib and the timing error of a matched pair is (|d start| + |d end|) / 2
(/root/reference/cli_benchmark.py:37-62 similarity, :64-108 word timing)."""
import difflib
import re
from typing import Dict, List, Optional, Sequence

import numpy as np


def _norm(word: str) -> str:
    return re.sub(r"[^\w']", "", word.lower())


def flatten_words(result: Dict) -> List[Dict]:
    """all words that carry times, in order, from a result dict {"segments": [{"words": [...]}]} (types.py:4-69)"""
    return [w for s in result.get("segments", []) for w in s.get("words", []) if "start" in w and "end" in w]


def word_mae_ms(got: Sequence[Dict], ref: Sequence[Dict]) -> Dict[str, Optional[float]]:
    """got / ref: lists of {"word", "start", "end"}.  Returns mae_ms, start_mae_ms, end_mae_ms, the share of reference
    words that found a partner, and the share of matched words within 20 ms (north_star tolerance)."""
    a, b = [_norm(w["word"]) for w in got], [_norm(w["word"]) for w in ref]
    sm = difflib.SequenceMatcher(a=a, b=b, autojunk=False)
    ds, de = [], []
    for blk in sm.get_matching_blocks():
        for k in range(blk.size):
            g, r = got[blk.a + k], ref[blk.b + k]
            ds.append(abs(float(g["start"]) - float(r["start"])))
            de.append(abs(float(g["end"]) - float(r["end"])))
    if not ds:
        return {"mae_ms": None, "start_mae_ms": None, "end_mae_ms": None, "matched": 0.0, "within_20ms": None}
    ds, de = np.asarray(ds), np.asarray(de)
    err = (ds + de) / 2.0
    return {"mae_ms": round(1e3 * float(err.mean()), 2), "start_mae_ms": round(1e3 * float(ds.mean()), 2),
            "end_mae_ms": round(1e3 * float(de.mean()), 2), "matched": round(len(ds) / max(1, len(ref)), 4),
            "within_20ms": round(float(((ds <= 0.020 + 1e-9) & (de <= 0.020 + 1e-9)).mean()), 4)}


def token_similarity(got: Sequence[int], ref: Sequence[int]) -> float:
    """difflib ratio of two token-id lists (1.0 = identical greedy ids)"""
    return difflib.SequenceMatcher(a=list(got), b=list(ref), autojunk=False).ratio()
