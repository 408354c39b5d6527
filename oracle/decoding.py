"""Oracle: batched greedy decode loop + logit filters.  TEST INFRASTRUCTURE ONLY.

Follows the reference's in-tree batch decoder:
  BatchGreedyDecoder.update      /root/reference/mlx_whisper_batch_decoder.py:267-303
  BatchDecodingTask._main_loop_batch                                        :317-384
  BatchDecodingTask.run (trim at EOT, avg_logprob)                          :386-468
  timestamp-probability rule (in-tree statement) /root/reference/mlx_ultra_optimized_batch.py:38-71
The remaining filters (SuppressBlank, SuppressTokens, full ApplyTimestampRules)
live in third-party mlx-whisper (PARITY UNPINNED, see oracle/__init__.py) and
are restated from the published OpenAI Whisper decoding rules; the timestamp
rules are cross-checked against transformers' WhisperTimeStampLogitsProcessor.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import whisper_ref as W

# rule bits (same values as include/wxhip.h WX_RULE_*)
RULE_SUPPRESS_BLANK = 1
RULE_SUPPRESS_TOKENS = 2
RULE_TS_NOTIMESTAMPS = 4     # -inf on <|notimestamps|>
RULE_TS_PAIRS = 8            # timestamps come in pairs
RULE_TS_MONOTONE = 16        # timestamps never decrease
RULE_TS_INITIAL = 32         # first sampled token is a timestamp <= max_initial
RULE_TS_PROB = 64            # sum p(timestamps) > max p(text) -> force timestamp
RULES_LIGHTNING = 127        # mlx_lightning.py:187-193 (DecodingOptions defaults)
RULES_OPTIMIZED_FINAL = RULE_SUPPRESS_TOKENS | RULE_TS_PROB   # optimized_final.py:301-306 + patched apply


@dataclass
class Specials:
    """Special-token layout of the multilingual Whisper vocabularies."""
    n_vocab: int
    eot: int = 50257
    sot: int = 50258
    n_langs: int = 99
    translate: int = 0
    transcribe: int = 0
    sot_lm: int = 0
    sot_prev: int = 0
    no_speech: int = 0
    no_timestamps: int = 0
    timestamp_begin: int = 0
    blank_tokens: List[int] = field(default_factory=lambda: [220])   # tokenizer.encode(" ")

    @staticmethod
    def for_vocab(n_vocab):
        n_langs = 100 if n_vocab >= 51866 else 99
        s = Specials(n_vocab=n_vocab, n_langs=n_langs)
        base = s.sot + 1 + n_langs
        s.translate, s.transcribe, s.sot_lm, s.sot_prev = base, base + 1, base + 2, base + 3
        s.no_speech, s.no_timestamps, s.timestamp_begin = base + 4, base + 5, base + 6
        return s

    def lang_token(self, idx=0):
        return self.sot + 1 + idx

    def initial_tokens(self, lang_idx=0, task="transcribe", without_timestamps=False):
        t = [self.sot, self.lang_token(lang_idx), self.transcribe if task == "transcribe" else self.translate]
        if without_timestamps:
            t.append(self.no_timestamps)
        return t


def apply_filters(logits, tokens, sp: Specials, sample_begin, rules, suppress_tokens=(),
                  max_initial_timestamp_index=50):
    """In-place logit filters on (B, vocab) fp32 given token history (B, n)."""
    B = logits.shape[0]
    NEG = float("-inf")
    n = tokens.shape[1]
    if (rules & RULE_SUPPRESS_BLANK) and n == sample_begin:
        logits[:, list(sp.blank_tokens) + [sp.eot]] = NEG
    if (rules & RULE_SUPPRESS_TOKENS) and len(suppress_tokens):
        logits[:, list(suppress_tokens)] = NEG
    if rules & RULE_TS_NOTIMESTAMPS:
        logits[:, sp.no_timestamps] = NEG
    tb = sp.timestamp_begin
    for k in range(B):
        seq = tokens[k, sample_begin:].tolist()
        last_ts = len(seq) >= 1 and seq[-1] >= tb
        pen_ts = len(seq) < 2 or seq[-2] >= tb
        if rules & RULE_TS_PAIRS:
            if last_ts:
                if pen_ts:
                    logits[k, tb:] = NEG
                else:
                    logits[k, : sp.eot] = NEG
        if rules & RULE_TS_MONOTONE:
            ts = [t for t in seq if t >= tb]
            if ts:
                last = ts[-1] if (last_ts and not pen_ts) else ts[-1] + 1
                logits[k, tb:last] = NEG
    if (rules & RULE_TS_INITIAL) and n == sample_begin:
        logits[:, :tb] = NEG
        if max_initial_timestamp_index is not None:
            logits[:, tb + max_initial_timestamp_index + 1:] = NEG
    if rules & RULE_TS_PROB:
        # mlx_ultra_optimized_batch.py:52-69
        logprobs = logits - torch.logsumexp(logits, dim=-1, keepdim=True)
        ts_lp = torch.logsumexp(logprobs[:, tb:], dim=-1)
        txt_lp = logprobs[:, :tb].max(dim=-1).values
        force = ts_lp > txt_lp
        logits[force, :tb] = NEG
    return logits


def greedy_update(tokens, logits, sum_logprobs, eot):
    """mlx_whisper_batch_decoder.py:267-303 at temperature 0."""
    nxt = logits.argmax(dim=-1)
    logprobs = logits - torch.logsumexp(logits, dim=-1, keepdim=True)
    cur = logprobs[torch.arange(logits.shape[0]), nxt]
    not_eot = tokens[:, -1] != eot
    sum_logprobs = sum_logprobs + torch.where(not_eot, cur, torch.zeros_like(cur))
    nxt = torch.where(tokens[:, -1] == eot, torch.full_like(nxt, eot), nxt)
    tokens = torch.cat([tokens, nxt[:, None]], dim=-1)
    return tokens, tokens[:, -1] == eot, sum_logprobs


@dataclass
class DecodeResult:
    tokens: List[List[int]]          # after sample_begin, cut at first EOT (run():433-441)
    raw_tokens: np.ndarray           # (B, n) everything incl. prompt
    sum_logprobs: np.ndarray
    avg_logprobs: List[float]        # sum / (len + 1)  (:446-449)
    no_speech_probs: np.ndarray
    cross_qk: Optional[list] = None  # per step: list over layers of (B,H,q,1500)
    step_logits: Optional[list] = None


@torch.no_grad()
def greedy_decode(w, dims, enc, sp: Specials, initial_tokens, rules=RULES_LIGHTNING,
                  suppress_tokens=(), sample_len=224, max_initial_timestamp_index=50,
                  forced_len=None, keep_qk=False, keep_logits=False):
    """_main_loop_batch (:317-384): first call consumes the whole prompt without a
    cache (:335), then one token per step; finished rows keep emitting EOT; stop
    when all complete (:357) or tokens exceed n_text_ctx (:367).
    `forced_len` (bench workload only): EOT is suppressed and the loop
    runs exactly that many sampled tokens."""
    B = enc.shape[0]
    xkv = W.cross_kv(w, dims, enc)
    tokens = torch.tensor(initial_tokens, dtype=torch.long)[None].repeat(B, 1)
    sample_begin = tokens.shape[1]
    sum_lp = torch.zeros(B)
    qks, all_logits = [], []
    cache = None
    no_speech = None
    n_steps = sample_len if forced_len is None else forced_len
    for i in range(n_steps):
        if i == 0:
            logits, cache, qk = W.decoder_forward(w, dims, tokens, xkv, None, 0)
        else:
            if forced_len is None and bool((tokens[:, -1] == sp.eot).all()):
                break
            if tokens.shape[-1] > dims.n_text_ctx:
                break
            logits, cache, qk = W.decoder_forward(w, dims, tokens[:, -1:], xkv, cache, tokens.shape[1] - 1)
        logits = logits[:, -1].float().clone()
        if keep_logits:
            all_logits.append(logits.clone())
        if forced_len is not None:
            logits[:, sp.eot] = float("-inf")
        apply_filters(logits, tokens, sp, sample_begin, rules, suppress_tokens, max_initial_timestamp_index)
        if i == 0:
            # :346-352 -- taken AFTER the filters in this variant
            no_speech = torch.softmax(logits, dim=-1)[:, sp.no_speech]
        if keep_qk:
            qks.append([q[:, :, -1:, :].clone() for q in qk])
        tokens, _done, sum_lp = greedy_update(tokens, logits, sum_lp, sp.eot)
    out_tokens, avg = [], []
    for b in range(B):
        t = tokens[b, sample_begin:].tolist()
        if sp.eot in t:
            t = t[: t.index(sp.eot)]
        out_tokens.append(t)
        avg.append(float(sum_lp[b]) / (len(t) + 1))
    return DecodeResult(out_tokens, tokens.numpy(), sum_lp.numpy(), avg, no_speech.numpy(),
                        qks if keep_qk else None, all_logits if keep_logits else None)
