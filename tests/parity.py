"""Strict greedy-token parity against the oracle (test infrastructure).

The GPU decodes free-running in fp16/fp32-accumulate; the oracle is fp32.  A free-running
comparison can only say "identical until the first near-tie".  `check_tokens_strict` instead
teacher-forces the ORACLE along the token sequence the GPU produced and checks EVERY sampled
step of EVERY row: the GPU's token must be the oracle's argmax after the logit filters, unless
the oracle's own margin between its argmax and the GPU's token is below `tol` (a genuine
near-tie under fp16 storage).  The summed log-probability the GPU reports is checked against
the oracle's log-probabilities of the same tokens.  A row is followed until its first EOT
(BatchGreedyDecoder.update pins EOT afterwards, mlx_whisper_batch_decoder.py:267-303).
"""
from dataclasses import dataclass
from typing import List

import numpy as np
import torch

from oracle import decoding as OD
from oracle import whisper_ref as OW


@dataclass
class StrictReport:
    steps_checked: int
    near_ties: int                 # steps where GPU != oracle argmax but inside `tol`
    rows_identical: int            # rows whose every checked step was the oracle's argmax
    max_lp_err: float              # max |sum_logprob_gpu - sum_logprob_oracle(tokens_gpu)|
    mismatches: List[tuple]        # (row, position, gpu token, oracle token, margin) beyond tol
    rows: int = 0
    max_near_tie_margin: float = 0.0   # the widest oracle margin among the near-ties (how "near" they really were)

    def line(self, name):
        pct = 100.0 * self.near_ties / max(1, self.steps_checked)
        return (f"{name}: steps_checked={self.steps_checked} near_ties={self.near_ties} ({pct:.2f} %, widest oracle margin "
                f"{self.max_near_tie_margin:.4f}) rows_identical={self.rows_identical}/{self.rows} max_sum_logprob_err={self.max_lp_err:.4f} "
                f"mismatches={len(self.mismatches)}")


@torch.no_grad()
def teacher_forced_logits(ck, dims, enc, tokens, rows_per_slice=4):
    """fp32 oracle logits at every position of `tokens` (B, n) int64 (one causal pass; equal to the
    step-by-step KV-cached loop up to fp32 rounding).  Rows are independent: they go through the oracle a few at a time,
    so that the cross K/V of a slice (2 GB per 4 rows at large-v3 depth) and its cross-attention scores bound the host
    memory, not the batch."""
    out = []
    for a in range(0, tokens.shape[0], rows_per_slice):
        xkv = OW.cross_kv(ck, dims, enc[a: a + rows_per_slice].float().cpu())
        logits, _, _ = OW.decoder_forward(ck, dims, tokens[a: a + rows_per_slice].long(), xkv)
        out.append(logits)
        del xkv
    return torch.cat(out)


@torch.no_grad()
def check_tokens_strict(ck, dims, enc, gpu_tokens, n_prompt, n_sampled, sp, rules, suppress_tokens=(),
                        forced_len=None, max_initial_ts=50, tol=1e-2, gpu_sum_logprob=None, lp_tol=None):
    """gpu_tokens: (B, >= n_prompt + n_sampled) ints (prompt + sampled, EOT-filled).  Returns a
    StrictReport; the caller asserts on it (`mismatches == []`)."""
    toks = torch.as_tensor(np.asarray(gpu_tokens)[:, : n_prompt + n_sampled].astype(np.int64))
    B = toks.shape[0]
    logits = teacher_forced_logits(ck, dims, enc, toks[:, :-1] if toks.shape[1] > n_prompt else toks)
    steps = near = 0
    widest = 0.0
    mism = []
    row_clean = [True] * B
    sum_lp = np.zeros(B, dtype=np.float64)
    alive = np.ones(B, dtype=bool)
    for s in range(n_sampled):
        n = n_prompt + s                        # index of the token sampled at this step
        lg = logits[:, n - 1].float().clone()
        if forced_len:
            lg[:, sp.eot] = float("-inf")
        OD.apply_filters(lg, toks[:, :n], sp, n_prompt, rules, suppress_tokens, max_initial_ts)
        lp = lg - torch.logsumexp(lg, dim=-1, keepdim=True)
        top = lg.argmax(dim=-1)
        for b in range(B):
            if not alive[b]:
                assert int(toks[b, n]) == sp.eot, ("row must stay EOT after its first EOT", b, n)
                continue
            got, ref = int(toks[b, n]), int(top[b])
            steps += 1
            sum_lp[b] += float(lp[b, got])
            if got != ref:
                row_clean[b] = False
                margin = float(lg[b, ref] - lg[b, got])
                if margin < tol:
                    near += 1
                    widest = max(widest, margin)
                else:
                    mism.append((b, n, got, ref, margin))
            if got == sp.eot:
                alive[b] = False
    max_lp_err = 0.0
    if gpu_sum_logprob is not None:
        g = np.asarray(gpu_sum_logprob, dtype=np.float64)
        err = np.abs(g - sum_lp)
        if lp_tol is not None:
            bound = lp_tol * np.maximum(1.0, np.abs(sum_lp))
            for b in range(B):
                if err[b] > bound[b]:
                    mism.append((b, -1, float(g[b]), float(sum_lp[b]), float(err[b])))
        max_lp_err = float(err.max())
    return StrictReport(steps, near, int(sum(row_clean)), max_lp_err, mism, rows=B, max_near_tie_margin=widest)


def assert_strict(rep: StrictReport, max_near_tie_frac=0.02, min_rows_identical_frac=0.0):
    """the bounds of a strict check; the figures go into pytest's end-of-run summary whatever the outcome (VERDICT r03 weak #3:
    a drift of the near-tie count must be visible in pytest.log)"""
    import os
    name = os.environ.get("PYTEST_CURRENT_TEST", "strict check").split("::")[-1].replace(" (call)", "")
    if not getattr(rep, "logged", False):
        log_report(name, rep)
    assert rep.mismatches == [], rep.mismatches[:8]
    assert rep.near_ties <= max(1, int(max_near_tie_frac * rep.steps_checked)), (rep.near_ties, rep.steps_checked)
    assert rep.rows_identical >= min_rows_identical_frac * rep.rows, (rep.rows_identical, rep.rows)


def log_report(name, rep: StrictReport):
    """the figures of a strict check into pytest's end-of-run summary (tests/conftest.py) and onto stdout"""
    from tests import conftest
    line = rep.line(name)
    rep.logged = True
    conftest.PARITY_LOG.append(line)
    print(line)
    return line
