"""CPU: the strict token checker itself (tests/parity.py) -- the oracle's own free-running decode must pass it with no
near-tie at all, and a corrupted token or log-probability must be reported."""
import numpy as np
import torch

from oracle import decoding as OD
from oracle import whisper_ref as OW
from tests import parity as PAR

DIMS = OW.Dims(80, 1500, 64, 2, 1, 51865, 448, 64, 2, 2)


def _setup(rules, forced=None):
    w = OW.random_weights(DIMS, seed=1, std=0.2, emb_std=0.1)
    sp = OD.Specials.for_vocab(DIMS.n_vocab)
    g = torch.Generator().manual_seed(0)
    enc = torch.randn(3, 1500, 64, generator=g)
    res = OD.greedy_decode(w, DIMS, enc, sp, sp.initial_tokens(), rules=rules, sample_len=12, forced_len=forced)
    return w, sp, enc, res


def test_oracle_decode_passes_strict_check():
    for rules, forced in ((OD.RULES_LIGHTNING, None), (0, 9)):
        w, sp, enc, res = _setup(rules, forced)
        n = res.raw_tokens.shape[1] - 3
        rep = PAR.check_tokens_strict(w, DIMS, enc, res.raw_tokens, 3, n, sp, rules, forced_len=forced,
                                      gpu_sum_logprob=res.sum_logprobs, lp_tol=1e-4)
        PAR.assert_strict(rep)
        assert rep.near_ties == 0 and rep.rows_identical == 3 and rep.max_lp_err < 1e-3
        assert rep.steps_checked == (3 * 9 if forced else rep.steps_checked) and rep.steps_checked >= 3


def test_strict_check_reports_a_wrong_token_and_a_wrong_logprob():
    w, sp, enc, res = _setup(0, 9)
    bad = res.raw_tokens.copy()
    bad[1, 3 + 4] = (bad[1, 3 + 4] + 17) % 50000
    rep = PAR.check_tokens_strict(w, DIMS, enc, bad, 3, 9, sp, 0, forced_len=9)
    assert [m[:2] for m in rep.mismatches][0] == (1, 7) and rep.rows_identical <= 2
    rep = PAR.check_tokens_strict(w, DIMS, enc, res.raw_tokens, 3, 9, sp, 0, forced_len=9,
                                  gpu_sum_logprob=res.sum_logprobs + np.array([0, 0, 5.0]), lp_tol=1e-3)
    assert rep.mismatches and rep.mismatches[0][0] == 2 and rep.mismatches[0][1] == -1
