"""CPU: the logit-filter restatement (oracle/decoding.py apply_filters) against REFERENCE-HELD
evidence: the token lists of /root/reference/30m.json (a whisper-large-v3 run of the reference pipeline, committed as
tests/golden/gold30m_windows.json by tools/make_gold_tokens.py).  Every gold token must be admissible under the filters
given its own gold history -- a rule that masked one could not have produced the reference's output.  This pins the
special-token layout (timestamp_begin 50365, 100 language tokens), SuppressBlank, SuppressTokens, the timestamp pair /
monotonicity / initial-timestamp rules; it cannot pin the arithmetic of the model itself (no weights ship)."""
import json
import os

import numpy as np
import torch

from oracle import decoding as OD
from whisperx_mlx_amd.tokenizer import get_tokenizer

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gold30m_windows.json")
N_VOCAB = 51866


def _windows():
    with open(GOLD) as f:
        g = json.load(f)
    return g, [w["tokens"] for w in g["windows"]]


def test_fixture_shape_and_token_layout():
    g, wins = _windows()
    tok = get_tokenizer(N_VOCAB)
    sp = OD.Specials.for_vocab(N_VOCAB)
    assert g["n_segments"] == 743 and len(wins) == 81
    assert tok.timestamp_begin == sp.timestamp_begin == g["timestamp_begin"] == 50365
    assert (tok.eot, tok.sot, tok.no_timestamps, tok.no_speech) == (sp.eot, sp.sot, sp.no_timestamps, sp.no_speech)
    n = sum(len(w) for w in wins)
    assert abs(n / 60.0 - 145.3) < 0.05                    # the bench's forced decode length per 30 s
    assert max(len(w) for w in wins) <= 224                # sample_len
    for w in wins:                                         # the reference's windows open and close with a timestamp
        assert w[0] >= 50365 and w[-1] >= 50365 and max(w) < 50365 + 1501
        assert not any(tok.eot <= t < 50365 for t in w)    # no special token is ever sampled
    sup = set(tok.suppress_tokens())
    assert not any(t in sup for w in wins for t in w)      # SuppressTokens never bans a token the reference emitted


def _check(rules, logit_of_gold):
    _g, wins = _windows()
    tok = get_tokenizer(N_VOCAB)
    sp = OD.Specials.for_vocab(N_VOCAB)
    sup = tok.suppress_tokens()
    prompt = sp.initial_tokens()
    P = len(prompt)
    L = max(len(w) for w in wins)
    n_checked = 0
    for s in range(L + 1):
        rows = [w for w in wins if len(w) >= s]            # step len(w) samples EOT after the closing timestamp
        hist = torch.tensor([prompt + w[:s] for w in rows], dtype=torch.long)
        gold = torch.tensor([w[s] if s < len(w) else sp.eot for w in rows], dtype=torch.long)
        lg = torch.zeros(len(rows), N_VOCAB)
        lg[torch.arange(len(rows)), gold] = logit_of_gold
        OD.apply_filters(lg, hist, sp, P, rules, sup, 50)
        kept = lg[torch.arange(len(rows)), gold]
        bad = torch.nonzero(~torch.isfinite(kept)).flatten().tolist()
        assert not bad, (s, [(rows[i][max(0, s - 3): s], int(gold[i])) for i in bad[:3]])
        if logit_of_gold > 0:
            assert (lg.argmax(dim=-1) == gold).all(), s     # and the greedy choice is the gold token
        n_checked += len(rows)
    return n_checked


def test_gold_tokens_admissible_under_structural_rules():
    # everything but the probability rule, which needs the model's logits: flat logits, the gold token must survive
    n = _check(OD.RULES_LIGHTNING & ~OD.RULE_TS_PROB, 0.0)
    assert n == 8716 + 81


def test_gold_tokens_chosen_under_all_rules():
    # all DecodingOptions-default rules (mlx_lightning.py:187-193) with the gold token well ahead of a flat field
    # (mass of the 1501 flat timestamps: log 1501 = 7.3 < 20, so the probability rule does not override a text token)
    assert _check(OD.RULES_LIGHTNING, 20.0) == 8716 + 81


def test_compression_ratio_matches_the_reference_run():
    """compression_ratio of a decode window (mlx_whisper_batch_decoder.py:470-477: utf-8 bytes / zlib-compressed bytes of
    the window's text) against the value the reference stored for each of its 81 windows"""
    from whisperx_mlx_amd.backend import _compression_ratio
    g, _ = _windows()
    for w in g["windows"]:
        assert abs(_compression_ratio(w["text"].strip()) - w["compression_ratio"]) < 1e-12, w["first_segment"]
