"""-m gpu: the benchmarked shape (BASELINE.json config 3: whisper-large-v3, d = 1280, 20 heads, 128 mel bins,
vocabulary 51866 with 100 language tokens, batch 16) against the CPU oracle.

Two models: (a) large-v3 WIDTH with 2 encoder + 2 decoder layers, every stage compared with the oracle at B = 16
(encoder, teacher-forced logits, sampler, greedy tokens step by step, alignment-head capture + DTW); (b) the full
32 + 32 layer model with the bench's seeded N(0, 0.02^2) weights: finite output, run-to-run identical tokens, clean
device status, plus one row of the full-depth encoder and its teacher-forced logits against the oracle.
Tolerances are fp16 storage / fp32 accumulation against an fp32 oracle and are written at each assert."""
import ctypes as C
import functools

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import gpu_util as G        # noqa: E402
from tests import parity as PAR        # noqa: E402
from oracle import decoding as OD      # noqa: E402
from oracle import dtw as ODTW         # noqa: E402
from oracle import whisper_ref as OW   # noqa: E402
from whisperx_mlx_amd import _lib      # noqa: E402
from whisperx_mlx_amd import engine as E   # noqa: E402
from whisperx_mlx_amd import weights as WT  # noqa: E402
from whisperx_mlx_amd.tokenizer import get_tokenizer  # noqa: E402

B = 16
WIDE = WT.ModelDimensions(128, 1500, 1280, 20, 2, 51866, 448, 1280, 20, 2)
WIDE_HEADS = [(0, 3), (0, 17), (1, 0), (1, 7), (1, 12), (1, 19)]
ENC_TOL = 2e-2          # rel to max |enc| (2 layers + conv stem, fp16 residual stream)
LOGIT_TOL = 3e-2        # abs, logits of O(1) spread
MARGIN_TOL = 1e-2     # a token that is not the oracle's argmax is tolerated only where the oracle's own margin is below this (widest seen over the whole suite: 0.0025)


@functools.lru_cache(maxsize=None)
def wide():
    # std ~ 1/sqrt(d): activations and logits of O(1), so that the random model emits text, timestamps and EOT
    ck = WT.random_checkpoint(WIDE, seed=3, std=0.03, emb_std=0.03)
    eng = E.WhisperHipEngine(WIDE, WT.pack(ck, WIDE, "cuda"), max_batch=B, alignment_heads=WIDE_HEADS)
    return eng, {k: v.float() for k, v in ck.items()}


def _mel(n, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(n, 3000, 128, generator=g) * 0.5).half()


def _oracle_encoder(ck, dims, mel, rows=4):
    """fp32 oracle in slices of `rows` chunks (the (rows, 20, 1500, 1500) score tensor is what bounds host memory)"""
    return torch.cat([OW.encoder_forward(ck, dims, mel[i: i + rows].float()) for i in range(0, mel.shape[0], rows)])


def test_wide_encoder_b16():
    eng, ck = wide()
    mel = _mel(B, 1)
    enc = eng.encode(mel.cuda())
    torch.cuda.synchronize()
    ref = _oracle_encoder(ck, WIDE, mel)
    assert torch.isfinite(enc).all()
    for b in range(B):                      # per chunk, so that one bad row cannot hide behind the batch maximum
        assert G.rel_err(enc[b], ref[b]) < ENC_TOL, b
    eng.check_status()


def test_capped_encoder_is_bit_identical():
    """wx_set_encoder_cap (lab hook, tools/ab_pipeline.py): the encoder's GEMM and attention launches as persistent blocks on
    at most N compute units -- the same tiles in the same arithmetic, so the same bits, whatever the cap"""
    eng, _ck = wide()
    mel = _mel(B, 31).cuda()
    ref = eng.encode(mel).clone()
    try:
        for cap in (8, 64, 200):
            eng.set_encoder_cap(cap)
            assert torch.equal(eng.encode(mel), ref), cap
        with pytest.raises(Exception):
            eng.set_encoder_cap(12)            # not a multiple of 8: the XCD-aware tile order would break
    finally:
        eng.set_encoder_cap(0)
    assert torch.equal(eng.encode(mel), ref)
    eng.check_status()


def test_tile_pipelined_gemms_equal_the_classic_kernels_through_encoder_and_cross_kv():
    """Every GEMM form of the hot path at its real shape -- Q | K projection with the attention scale folded into its Q half,
    out-proj + residual in place, FC1 + GELU, FC2 + residual, and the cross-K/V projection's head-split layout -- through
    gemm_pipe_kernel (straight-line epilogue on interior tiles, the general one on the ragged edge: 16 x 1500 rows are 93.75
    tiles) against the one-tile-per-block kernel (wx_set_encoder_cap(-1)): encoder output and teacher-forced logits equal to
    the bit.  (The straight-line epilogue first differed by one fp16 ulp in ~7e-5 of the GELU outputs: the compiler had folded
    the GELU's last fma and the conversion into one v_fma_mixlo_f16 -- one rounding instead of two.)"""
    eng, _ck = wide()
    mel = _mel(B, 32).cuda()
    tok = get_tokenizer(WIDE.n_vocab)
    g = torch.Generator().manual_seed(6)
    toks = torch.cat([torch.tensor(tok.sot_sequence())[None].repeat(B, 1), torch.randint(0, 50000, (B, 5), generator=g)], 1)
    try:
        eng.set_encoder_cap(-1)
        enc_ref = eng.encode(mel).clone()
        lg_ref = eng.decode_logits(enc_ref, toks).clone()
        eng.set_encoder_cap(0)
        enc = eng.encode(mel)
        assert torch.equal(enc, enc_ref)
        assert torch.equal(eng.decode_logits(enc, toks), lg_ref)
    finally:
        eng.set_encoder_cap(0)
    eng.check_status()


def test_wide_logits_teacher_forced_b16():
    eng, ck = wide()
    enc = eng.encode(_mel(B, 2).cuda())
    tok = get_tokenizer(WIDE.n_vocab)
    assert tok.timestamp_begin == 50365 and tok.n_langs == 100      # pinned by /root/reference/30m.json (tests/golden)
    g = torch.Generator().manual_seed(5)
    full = torch.cat([torch.tensor(tok.sot_sequence())[None].repeat(B, 1), torch.randint(0, 50000, (B, 12), generator=g)], 1)
    xkv = OW.cross_kv(ck, WIDE, enc.float().cpu())
    for n in (1, 3, 11):
        toks = full[:, :n]
        lg = eng.decode_logits(enc, toks)
        torch.cuda.synchronize()
        ref, _, _ = OW.decoder_forward(ck, WIDE, toks.long(), xkv)
        err = (lg.cpu() - ref[:, -1]).abs().amax(dim=1)
        assert float(err.max()) < LOGIT_TOL, (n, err.tolist())
    eng.check_status()


@pytest.mark.parametrize("rules", [OD.RULES_LIGHTNING, OD.RULES_OPTIMIZED_FINAL])
def test_wide_sampler_exact_on_oracle_logits(rules):
    """51866-entry vocabulary (100 language tokens, timestamp_begin 50365), 16 rows: fed the oracle's raw logits and
    history the kernel must choose the oracle's token at every step (integer parity)."""
    eng, ck = wide()
    tok = get_tokenizer(WIDE.n_vocab)
    sp = OD.Specials.for_vocab(WIDE.n_vocab)
    enc = eng.encode(_mel(B, 3).cuda()).float().cpu()
    sup = tok.suppress_tokens()
    res = OD.greedy_decode(ck, WIDE, enc, sp, tok.sot_sequence(), rules=rules, suppress_tokens=sup, sample_len=24,
                           keep_logits=True)
    ids = set()
    if rules & OD.RULE_SUPPRESS_TOKENS:
        ids.update(sup)
    if rules & OD.RULE_TS_NOTIMESTAMPS:
        ids.add(tok.no_timestamps)
    mask = eng.suppress_mask(sorted(ids))
    o = _lib.DecodeOpts()
    P = len(tok.sot_sequence())
    for i, t in enumerate(tok.sot_sequence()):
        o.prompt[i] = t
    o.n_prompt, o.sample_len, o.rules, o.max_initial_ts = P, 224, rules, 50
    o.eot, o.no_speech, o.timestamp_begin = tok.eot, tok.no_speech, tok.timestamp_begin
    o.blank0, o.blank1 = tok.blank_tokens[0], -1
    o.suppress_mask = mask.data_ptr()
    raw = torch.from_numpy(res.raw_tokens).int()
    tokens = torch.full((B, WIDE.n_text_ctx), tok.eot, dtype=torch.int32)
    tokens[:, :P] = raw[:, :P]
    tokens = tokens.cuda()
    slp = torch.zeros(B, device="cuda")
    nsp = torch.zeros(B, device="cuda")
    L = _lib.lib()
    for i, lg in enumerate(res.step_logits):
        n = P + i
        tokens[:, :n] = raw[:, :n].cuda()
        ld = (WIDE.n_vocab + 3) // 4 * 4 if i % 2 else WIDE.n_vocab
        lgd = torch.zeros(B, ld, device="cuda")
        lgd[:, : WIDE.n_vocab] = lg.float().cuda()
        rc = L.wx_sample_step(eng.ctx, _lib.ptr(lgd), lgd.stride(0), _lib.ptr(tokens), WIDE.n_text_ctx, n, B,
                              C.byref(o), _lib.ptr(slp), _lib.ptr(nsp), None)
        _lib.check(eng.ctx, rc, "wx_sample_step")
        torch.cuda.synchronize()
        assert tokens[:, n].cpu().tolist() == raw[:, n].tolist(), (i, rules)
    assert np.allclose(slp.cpu().numpy(), res.sum_logprobs, rtol=1e-4, atol=1e-3)
    assert np.allclose(nsp.cpu().numpy(), res.no_speech_probs, atol=1e-6)


@pytest.mark.parametrize("split,forced", [(2, 0), (1, 0), (2, 40)])
def test_wide_greedy_tokens_strict_b16(split, forced):
    """the device-resident greedy loop (hipGraph, key splits) at the bench's width and batch: every step of every row"""
    eng, ck = wide()
    tok = get_tokenizer(WIDE.n_vocab)
    sp = OD.Specials.for_vocab(WIDE.n_vocab)
    enc = eng.encode(_mel(B, 4).cuda())
    rules = 0 if forced else E.RULES_LIGHTNING
    out = eng.decode(enc, tok, tok.sot_sequence(), rules=rules, suppress_ids=tok.suppress_tokens(), sample_len=40,
                     forced_len=forced, cross_split=split)
    eng.check_status()
    rep = PAR.check_tokens_strict(ck, WIDE, enc, out.tokens.cpu().numpy(), out.n_prompt, out.n_sampled, sp,
                                  OD.RULES_LIGHTNING if not forced else 0, tok.suppress_tokens(),
                                  forced_len=forced or None, tol=MARGIN_TOL,
                                  gpu_sum_logprob=out.sum_logprob.cpu().numpy(), lp_tol=0.01)
    PAR.assert_strict(rep)
    assert rep.steps_checked >= (B * 40 if forced else 40)
    assert rep.rows_identical >= 1
    # the same decode again: bit-identical tokens and log-probabilities (fixed summation order, no float atomics)
    t1, l1 = out.tokens.cpu().numpy().copy(), out.sum_logprob.cpu().numpy().copy()
    out2 = eng.decode(enc, tok, tok.sot_sequence(), rules=rules, suppress_ids=tok.suppress_tokens(), sample_len=40,
                      forced_len=forced, cross_split=split)
    assert np.array_equal(out2.tokens.cpu().numpy(), t1) and np.array_equal(out2.sum_logprob.cpu().numpy(), l1)


def test_wide_capture_and_dtw_b16():
    eng, ck = wide()
    tok = get_tokenizer(WIDE.n_vocab)
    enc = eng.encode(_mel(B, 6).cuda())
    out = eng.decode(enc, tok, tok.sot_sequence(), rules=E.RULES_LIGHTNING, suppress_ids=tok.suppress_tokens(),
                     sample_len=20, capture_qk=True)
    qk = eng.align_qk(B).cpu().numpy()
    toks = out.tokens.cpu().numpy()
    P, S = out.n_prompt, out.n_sampled
    xkv = OW.cross_kv(ck, WIDE, enc.float().cpu())
    for b in (0, 7, 15):
        seq = torch.from_numpy(toks[b: b + 1, : P + S].astype(np.int64))
        _, _, cq = OW.decoder_forward(ck, WIDE, seq, [(k[b: b + 1], v[b: b + 1]) for k, v in xkv])
        for hi, (l, h) in enumerate(WIDE_HEADS):
            ref = cq[l][0, h, P - 1: P - 1 + S].numpy()
            assert np.abs(qk[b, hi, :S] - ref).max() < 5e-3 * np.abs(ref).max() + 1e-2, (b, l, h)
    paths, mat = eng.dtw_path(out, tok.eot, mode=0, want_matrix=True)
    mat = mat.cpu().numpy()
    eng.check_status()
    for b in range(B):
        rows = []
        for s, t in enumerate(toks[b, P: P + S].tolist()):
            if t <= tok.eot:
                rows.append(s)
            if t == tok.eot:
                break
        n_rows, path = paths[b]
        assert n_rows == len(rows)
        if n_rows < 2:
            continue
        ref_m = ODTW.alignment_matrix_upstream(qk[b][:, rows, :])
        assert np.abs(mat[b, :n_rows] - ref_m).max() < 5e-3, b
        assert np.array_equal(path, ODTW.dtw_path_fast(-mat[b, :n_rows])), b      # integer parity of the DP


# ------------------------------------------------------------------------------- full depth (32 + 32 layers)
@functools.lru_cache(maxsize=None)
def full():
    dims = WT.MODEL_DIMS["large-v3"]
    ck = WT.random_checkpoint(dims, seed=0, std=0.02, device="cuda")        # bench.py's weights
    heads = WT.default_alignment_heads("large-v3", dims)
    eng = E.WhisperHipEngine(dims, WT.pack(ck, dims, "cuda"), max_batch=B, alignment_heads=heads)
    return eng, ck, dims


def test_full_large_v3_b16_finite_deterministic():
    eng, _ck, dims = full()
    tok = get_tokenizer(dims.n_vocab)
    mel = _mel(B, 8)
    enc = eng.encode(mel.cuda())
    assert torch.isfinite(enc).all()
    e1 = enc.clone()
    enc = eng.encode(mel.cuda())
    assert torch.equal(enc, e1)                                   # encoder is run-to-run bit-identical
    kw = dict(rules=0, forced_len=48, capture_qk=True)
    a = eng.decode(enc, tok, tok.sot_sequence(), **kw)
    ta, la = a.tokens.cpu().numpy().copy(), a.sum_logprob.cpu().numpy().copy()
    pa = eng.dtw_path(a, tok.eot)
    eng.check_status()
    assert a.n_sampled == 48 and np.isfinite(la).all() and (la < 0).all()
    P = a.n_prompt
    assert ((ta[:, P: P + 48] >= 0) & (ta[:, P: P + 48] < dims.n_vocab) & (ta[:, P: P + 48] != tok.eot)).all()
    b = eng.decode(enc, tok, tok.sot_sequence(), **kw)
    assert np.array_equal(b.tokens.cpu().numpy(), ta) and np.array_equal(b.sum_logprob.cpu().numpy(), la)
    pb = eng.dtw_path(b, tok.eot)
    for r, ((na, xa), (nb, xb)) in enumerate(zip(pa, pb)):
        n_text = int((ta[r, P: P + 48] < tok.eot).sum())          # rows of the alignment matrix = text tokens
        assert na == nb == n_text and np.array_equal(xa, xb)
        assert xa[0, 0] == 0 and xa[1, 0] == 0 and xa[0, -1] == n_text - 1 and xa[1, -1] == 1499
        assert (np.diff(xa[0]) >= 0).all() and (np.diff(xa[1]) >= 0).all()
    # rows do not depend on their batch position or on the batch size
    c = eng.decode(enc[5:9].contiguous(), tok, tok.sot_sequence(), rules=0, forced_len=48)
    assert np.array_equal(c.tokens.cpu().numpy()[:4], ta[5:9])
    # one kernel per stage (step variant 1) gives the bits of the fused launches (the default)
    v1 = eng.decode(enc, tok, tok.sot_sequence(), step_variant=1, **kw)
    assert np.array_equal(v1.tokens.cpu().numpy(), ta) and np.array_equal(v1.sum_logprob.cpu().numpy(), la)
    # free-running with the default filters: terminates cleanly whatever the random model emits
    d = eng.decode(enc, tok, tok.sot_sequence(), rules=E.RULES_LIGHTNING, suppress_ids=tok.suppress_tokens())
    eng.check_status()
    td = d.tokens.cpu().numpy()
    assert (td[:, P] >= tok.timestamp_begin).all() and (td[:, P] <= tok.timestamp_begin + 50).all()


def test_full_large_v3_one_row_against_oracle():
    """32 encoder layers and 32 decoder layers of one chunk against the fp32 oracle (about a minute of CPU time):
    the residual stream is fp16 on the GPU, so the bounds are looser than for 2 layers and stated here."""
    eng, ck, dims = full()
    ck32 = {k: v.float().cpu() for k, v in ck.items()}
    tok = get_tokenizer(dims.n_vocab)
    mel = _mel(B, 9)
    enc = eng.encode(mel.cuda())
    ref = OW.encoder_forward(ck32, dims, mel[3:4].float())
    rel = G.rel_err(enc[3], ref[0])
    assert rel < 4e-2, rel
    toks = torch.tensor([list(tok.sot_sequence()) + [50365, 2425, 11, 257, 1002]] * B, dtype=torch.int32)
    lg = eng.decode_logits(enc, toks.cuda()).cpu()
    eng.check_status()
    lg_ref, _, _ = OW.decoder_forward(ck32, dims, toks[3:4].long(), OW.cross_kv(ck32, dims, enc[3:4].float().cpu()))
    lg_ref = lg_ref[0, -1]
    err = float((lg[3] - lg_ref).abs().max())
    assert err < 5e-2 * max(1.0, float(lg_ref.abs().max())), err
    assert int(lg[3].argmax()) == int(lg_ref.argmax()) or float(lg_ref.max() - lg_ref[int(lg[3].argmax())]) < MARGIN_TOL


def test_full_large_v3_three_rows_strict_greedy_against_oracle():
    """The benchmarked model itself -- 32 + 32 layers, the bench's weights, batch 16, the default logit filters -- under
    the strict token check for the first, a middle and the last row of the batch: the oracle (fp32, 32 decoder layers) is
    teacher-forced along the 20 tokens the GPU sampled for each of them and every step must be its argmax after the
    filters (near-ties below the stated margin are counted and bounded); the summed log-probabilities agree within 1 %.
    The rows' encoder outputs are checked against the oracle's 32-layer encoder too (first and last row)."""
    eng, ck, dims = full()
    ck32 = {k: v.float().cpu() for k, v in ck.items()}
    tok = get_tokenizer(dims.n_vocab)
    sp = OD.Specials.for_vocab(dims.n_vocab)
    mel = _mel(B, 21)
    enc = eng.encode(mel.cuda())
    rows = [0, 7, B - 1]
    for r in (0, B - 1):
        ref = OW.encoder_forward(ck32, dims, mel[r: r + 1].float())
        assert G.rel_err(enc[r], ref[0]) < 4e-2, r
    out = eng.decode(enc, tok, tok.sot_sequence(), rules=E.RULES_LIGHTNING, suppress_ids=tok.suppress_tokens(), sample_len=20)
    eng.check_status()
    toks, slp = out.tokens.cpu().numpy(), out.sum_logprob.cpu().numpy()
    rep = PAR.check_tokens_strict(ck32, dims, enc[rows], toks[rows], out.n_prompt, out.n_sampled, sp, OD.RULES_LIGHTNING,
                                  tok.suppress_tokens(), tol=MARGIN_TOL, gpu_sum_logprob=slp[rows], lp_tol=0.01)
    PAR.assert_strict(rep)
    assert rep.steps_checked >= 3 * 16


def test_sampler_never_masks_a_reference_token():
    """sample.hip against REFERENCE-HELD evidence: the 81 decode windows of /root/reference/30m.json
    (tests/golden/gold30m_windows.json).  Teacher-forced along each window's own history, with the gold token 20 ahead
    of a flat field, the kernel must pick the gold token at every one of the 8716 steps under the default rule set
    (a filter that masked it could not have produced the reference's output), then EOT, then stay at EOT."""
    import json
    import os
    eng, _ = wide()
    tok = get_tokenizer(WIDE.n_vocab)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gold30m_windows.json")) as f:
        wins = [w["tokens"] for w in json.load(f)["windows"]]
    prompt = tok.sot_sequence("en", "transcribe")
    P = len(prompt)
    mask = eng.suppress_mask(sorted(set(tok.suppress_tokens()) | {tok.no_timestamps}))
    o = _lib.DecodeOpts()
    for i, t in enumerate(prompt):
        o.prompt[i] = t
    o.n_prompt, o.sample_len, o.rules, o.max_initial_ts = P, 224, E.RULES_LIGHTNING, 50
    o.eot, o.no_speech, o.timestamp_begin = tok.eot, tok.no_speech, tok.timestamp_begin
    o.blank0, o.blank1 = tok.blank_tokens[0], -1
    o.suppress_mask = mask.data_ptr()
    L = _lib.lib()
    n_steps = 0
    for g0 in range(0, len(wins), B):
        grp = wins[g0: g0 + B]
        n = len(grp)
        gold = np.full((n, 226), tok.eot, dtype=np.int64)
        for r, w in enumerate(grp):
            gold[r, : len(w)] = w
        tokens = torch.full((n, WIDE.n_text_ctx), tok.eot, dtype=torch.int32)
        tokens[:, :P] = torch.tensor(prompt, dtype=torch.int32)
        tokens = tokens.cuda()
        slp = torch.zeros(n, device="cuda")
        nsp = torch.zeros(n, device="cuda")
        rows = torch.arange(n, device="cuda")
        for s in range(max(len(w) for w in grp) + 2):
            lg = torch.zeros(n, WIDE.n_vocab, device="cuda")
            lg[rows, torch.from_numpy(gold[:, s]).cuda()] = 20.0
            rc = L.wx_sample_step(eng.ctx, _lib.ptr(lg), lg.stride(0), _lib.ptr(tokens), WIDE.n_text_ctx, P + s, n,
                                  C.byref(o), _lib.ptr(slp), _lib.ptr(nsp), None)
            _lib.check(eng.ctx, rc, "wx_sample_step")
            got = tokens[:, P + s].cpu().numpy()
            assert np.array_equal(got, gold[:, s]), (g0, s, got.tolist(), gold[:, s].tolist())
            n_steps += int((gold[:, s] != tok.eot).sum())
    assert n_steps == 8716


@pytest.mark.parametrize("d,H", [(768, 12), (1024, 16)])
def test_other_model_widths_small_and_medium(d, H):
    """whisper-small / -medium width (12 / 16 heads) with one encoder and two decoder layers: encoder output, teacher-forced
    logits and strict greedy steps against the oracle -- the GEMV k-splits (24 / 32 k-steps over 8 waves), the 256-tile GEMM
    with 3 / 4 column tiles and the fused launch at head counts the other tests do not visit"""
    dims = WT.ModelDimensions(80, 1500, d, H, 1, 51865, 448, d, H, 2)
    ck = WT.random_checkpoint(dims, seed=d, std=0.035, emb_std=0.035)
    eng = E.WhisperHipEngine(dims, WT.pack(ck, dims, "cuda"), max_batch=5, alignment_heads=[(1, 0), (1, H - 1)])
    ck32 = {k: v.float() for k, v in ck.items()}
    tok = get_tokenizer(dims.n_vocab)
    sp = OD.Specials.for_vocab(dims.n_vocab)
    g = torch.Generator().manual_seed(d)
    mel = (torch.randn(5, 3000, 80, generator=g) * 0.5).half()
    enc = eng.encode(mel.cuda())
    ref = OW.encoder_forward(ck32, dims, mel.float())
    assert G.rel_err(enc, ref) < ENC_TOL
    out = eng.decode(enc, tok, tok.sot_sequence(), rules=E.RULES_LIGHTNING, suppress_ids=tok.suppress_tokens(), sample_len=24,
                     capture_qk=True)
    eng.check_status()
    rep = PAR.check_tokens_strict(ck32, dims, enc, out.tokens.cpu().numpy(), out.n_prompt, out.n_sampled, sp, OD.RULES_LIGHTNING,
                                  tok.suppress_tokens(), tol=MARGIN_TOL, gpu_sum_logprob=out.sum_logprob.cpu().numpy(), lp_tol=0.01)
    PAR.assert_strict(rep)
    t0 = out.tokens.cpu().numpy().copy()                  # (DecodeOutput views the engine's buffers)
    v1 = eng.decode(enc, tok, tok.sot_sequence(), rules=E.RULES_LIGHTNING, suppress_ids=tok.suppress_tokens(), sample_len=24,
                    capture_qk=True, step_variant=1)
    assert np.array_equal(v1.tokens.cpu().numpy(), t0)   # one kernel per stage = the fused launch, at this head count too
    paths = eng.dtw_path(v1, tok.eot)
    assert len(paths) == 5


def test_int8_fused_launch_equals_one_kernel_per_stage():
    """int8 decoder weights at large-v3 width: the fused decode launch (its GEMV role dequantises, scales and biases the
    query itself) against one kernel per stage -- tokens AND log-probabilities bit for bit, on the model and input where
    the two once differed in the last bit at position 9 (the compiler had folded "fma, then convert to fp16" into one
    v_fma_mix*_f16 in one copy of the epilogue and not in the other: one rounding against two; decode_dev.h scale_bias)"""
    one = WT.ModelDimensions(128, 1500, 1280, 20, 1, 51866, 448, 1280, 20, 1)
    ck = WT.random_checkpoint(one, seed=5, std=0.03, emb_std=0.03)
    eng = E.WhisperHipEngine(one, WT.quantize_packed_decoder(WT.pack(ck, one, "cuda"), one), max_batch=16)      # one decoder layer: it is int8
    tok = get_tokenizer(one.n_vocab)
    enc = eng.encode((torch.randn(16, 3000, one.n_mels, generator=torch.Generator().manual_seed(3)) * 0.5).half().cuda())
    kw = dict(rules=0, forced_len=24)
    a = eng.decode(enc, tok, tok.sot_sequence(), step_variant=1, **kw)
    ta, la = a.tokens.cpu().numpy().copy(), a.sum_logprob.cpu().numpy().copy()
    for use_graph in (True, False):
        b = eng.decode(enc, tok, tok.sot_sequence(), step_variant=0, use_graph=use_graph, **kw)
        eng.check_status()
        assert np.array_equal(b.tokens.cpu().numpy(), ta) and np.array_equal(b.sum_logprob.cpu().numpy(), la)
    eng.close()


def test_wide_int8_decoder_weights_strict_b16():
    """compute_type int8 at large-v3 width and batch 16: int8 decode GEMV weights + row scales (last layer fp16) against the
    oracle's fp32 decoder on the quantise -> dequantise checkpoint, every step of every row"""
    from oracle import quant as OQ
    ck = WT.random_checkpoint(WIDE, seed=3, std=0.03, emb_std=0.03)
    packed = WT.quantize_packed_decoder(WT.pack(ck, WIDE, "cuda"), WIDE)
    eng = E.WhisperHipEngine(WIDE, packed, max_batch=B, alignment_heads=WIDE_HEADS)
    ckq = OQ.dequantized_checkpoint({k: v.float() for k, v in ck.items()}, WIDE.n_text_layer)
    tok = get_tokenizer(WIDE.n_vocab)
    sp = OD.Specials.for_vocab(WIDE.n_vocab)
    enc = eng.encode(_mel(B, 14).cuda())
    out = eng.decode(enc, tok, tok.sot_sequence(), rules=E.RULES_LIGHTNING, suppress_ids=tok.suppress_tokens(), sample_len=32)
    eng.check_status()
    rep = PAR.check_tokens_strict(ckq, WIDE, enc, out.tokens.cpu().numpy(), out.n_prompt, out.n_sampled, sp, OD.RULES_LIGHTNING,
                                  tok.suppress_tokens(), tol=MARGIN_TOL, gpu_sum_logprob=out.sum_logprob.cpu().numpy(), lp_tol=0.01)
    PAR.assert_strict(rep)
    assert rep.steps_checked >= 32
