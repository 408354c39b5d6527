"""-m gpu: the Whisper hot path (encoder, decoder step, greedy loop with filters,
cross-attention DTW) through the C ABI against the CPU oracle on the same seeded
weights and inputs.  fp16 storage / fp32 accumulation on the GPU vs fp32 oracle:
tolerances are stated per test; token ids must match wherever the oracle's own
margin is above the floating tolerance."""
import ctypes as C

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
from whisperx_mlx_amd.tokenizer import get_tokenizer  # noqa: E402

DIMS = G.TEST_DIMS
ENC_TOL = 2e-2        # rel to max |enc|, fp16 residual stream over 2 layers + conv stem
LOGIT_TOL = 3e-2      # abs on logits of O(3) magnitude
MARGIN_TOL = 1e-2     # a token that is not the oracle's argmax is tolerated only where the oracle's own margin is below this (widest seen over the whole suite: 0.0025)


def _mel(B, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, 3000, DIMS.n_mels, generator=g) * 0.5).half()


def tiny():
    return G.tiny_engine()


def test_encoder_stem_and_layers():
    eng, ck = tiny()
    mel = _mel(3)
    enc = eng.encode(mel.cuda())
    torch.cuda.synchronize()
    ref = OW.encoder_forward(ck, DIMS, mel.float())
    assert torch.isfinite(enc).all()
    assert G.rel_err(enc, ref) < ENC_TOL


def test_decoder_logits_teacher_forced():
    eng, ck = tiny()
    mel = _mel(3, seed=1)
    enc = eng.encode(mel.cuda())
    ref_enc = enc.float().cpu()             # same encoder output on both sides: isolates the decoder
    tok = get_tokenizer(DIMS.n_vocab)
    g = torch.Generator().manual_seed(5)
    for n in (1, 3, 9):
        toks = torch.cat([torch.tensor(tok.sot_sequence())[None].repeat(3, 1),
                          torch.randint(0, 50000, (3, 16), generator=g)], 1)[:, :n]
        lg = eng.decode_logits(enc, toks)
        torch.cuda.synchronize()
        xkv = OW.cross_kv(ck, DIMS, ref_enc)
        ref, _, _ = OW.decoder_forward(ck, DIMS, toks.long(), xkv)
        assert (lg.cpu() - ref[:, -1]).abs().max().item() < LOGIT_TOL, n


def _opts(tok, rules, suppress, prompt, mask, forced=0):
    o = _lib.DecodeOpts()
    for i, t in enumerate(prompt):
        o.prompt[i] = t
    o.n_prompt = len(prompt)
    o.sample_len = 224
    o.rules = rules
    o.max_initial_ts = 50
    o.forced_len = forced
    o.eot, o.no_speech, o.timestamp_begin = tok.eot, tok.no_speech, tok.timestamp_begin
    o.blank0, o.blank1 = tok.blank_tokens[0], -1
    o.suppress_mask = mask.data_ptr()
    return o


@pytest.mark.parametrize("rules", [OD.RULES_LIGHTNING, OD.RULES_OPTIMIZED_FINAL, 0])
def test_sampler_kernel_exact_on_oracle_logits(rules):
    """Filters + greedy update in isolation: fed the oracle's raw logits and token
    history, the kernel must pick the oracle's token at every step (integer parity)."""
    eng, ck = tiny()
    tok = get_tokenizer(DIMS.n_vocab)
    sp = OD.Specials.for_vocab(DIMS.n_vocab)
    mel = _mel(3, seed=2)
    enc = OW.encoder_forward(ck, DIMS, mel.float())
    sup = tok.suppress_tokens()
    res = OD.greedy_decode(ck, DIMS, enc, sp, tok.sot_sequence(), rules=rules, suppress_tokens=sup, sample_len=48,
                           keep_logits=True)
    ids = set()
    if rules & OD.RULE_SUPPRESS_TOKENS:
        ids.update(sup)
    if rules & OD.RULE_TS_NOTIMESTAMPS:
        ids.add(tok.no_timestamps)
    mask = eng.suppress_mask(sorted(ids))
    o = _opts(tok, rules, sup, tok.sot_sequence(), mask)
    B, P = 3, len(tok.sot_sequence())
    raw = torch.from_numpy(res.raw_tokens).int()
    tokens = torch.full((B, DIMS.n_text_ctx), tok.eot, dtype=torch.int32)
    tokens[:, :P] = raw[:, :P]
    tokens = tokens.cuda()
    slp = torch.zeros(B, device="cuda")
    nsp = torch.zeros(B, device="cuda")
    L = _lib.lib()
    for i, lg in enumerate(res.step_logits):
        n = P + i
        tokens[:, :n] = raw[:, :n].cuda()          # teacher-force the oracle history
        # odd steps use a 16-byte aligned row stride (vectorised pass), even steps the exact vocab stride (scalar pass)
        ld = (DIMS.n_vocab + 3) // 4 * 4 if i % 2 else DIMS.n_vocab
        lgd = torch.zeros(B, ld, device="cuda")
        lgd[:, : DIMS.n_vocab] = lg.float().cuda()
        rc = L.wx_sample_step(eng.ctx, _lib.ptr(lgd), lgd.stride(0), _lib.ptr(tokens), DIMS.n_text_ctx, n, B,
                              C.byref(o), _lib.ptr(slp), _lib.ptr(nsp), None)
        _lib.check(eng.ctx, rc, "wx_sample_step")
        torch.cuda.synchronize()
        assert tokens[:, n].cpu().tolist() == raw[:, n].tolist(), (i, rules)
    assert np.allclose(slp.cpu().numpy(), res.sum_logprobs, rtol=1e-4, atol=1e-3)
    assert np.allclose(nsp.cpu().numpy(), res.no_speech_probs, atol=1e-6)


def _strict(ck, dims, enc, out, tok, rules, suppress=(), forced_len=None, lp_tol=0.01):
    """every sampled step of every row against the oracle teacher-forced along the GPU's own tokens
    (tests/parity.py): the GPU token is the oracle's argmax wherever the oracle's margin is >= MARGIN_TOL"""
    sp = OD.Specials.for_vocab(dims.n_vocab)
    rep = PAR.check_tokens_strict(ck, dims, enc, out.tokens.cpu().numpy(), out.n_prompt, out.n_sampled, sp, rules,
                                  suppress, forced_len=forced_len, tol=MARGIN_TOL,
                                  gpu_sum_logprob=out.sum_logprob.cpu().numpy(), lp_tol=lp_tol)
    PAR.assert_strict(rep)
    return rep


@pytest.mark.parametrize("use_graph,split", [(False, 1), (True, 4), (True, 2)])
def test_greedy_decode_tokens(use_graph, split):
    eng, ck = tiny()
    tok = get_tokenizer(DIMS.n_vocab)
    sp = OD.Specials.for_vocab(DIMS.n_vocab)
    mel = _mel(4, seed=3)
    enc = eng.encode(mel.cuda())
    out = eng.decode(enc, tok, tok.sot_sequence(), rules=E.RULES_LIGHTNING, suppress_ids=tok.suppress_tokens(),
                     sample_len=40, use_graph=use_graph, cross_split=split, check_every=8)
    torch.cuda.synchronize()
    eng.check_status()
    rep = _strict(ck, DIMS, enc, out, tok, OD.RULES_LIGHTNING, tok.suppress_tokens())
    assert rep.steps_checked >= 40          # the rows are not all finished after a handful of tokens
    # the oracle's own free-running decode: rows that never met a near-tie must be identical end to end
    res = OD.greedy_decode(ck, DIMS, enc.float().cpu(), sp, tok.sot_sequence(), rules=OD.RULES_LIGHTNING,
                           suppress_tokens=tok.suppress_tokens(), sample_len=40)
    got = out.tokens.cpu().numpy()
    n_same = sum(np.array_equal(got[b, : res.raw_tokens.shape[1]], res.raw_tokens[b]) for b in range(4))
    assert n_same >= rep.rows_identical >= 1


def test_greedy_decode_coalesced_requests():
    """37 chunks decoded in ONE launch (the 16-row GEMV kernels over three row groups) give exactly the rows that
    16-row requests decoded one after the other give: batch rows are independent.  The M-tiled kernels (decode
    step variant 3) keep the per-element summation order too."""
    eng, ck = G.tiny_engine(max_batch=40)
    tok = get_tokenizer(DIMS.n_vocab)
    mel = _mel(37, seed=11)
    enc = eng.encode(mel.cuda())
    kw = dict(rules=E.RULES_LIGHTNING, suppress_ids=tok.suppress_tokens(), sample_len=32, check_every=8)
    big = eng.decode(enc, tok, tok.sot_sequence(), **kw)
    tb, lb = big.tokens.cpu().numpy().copy(), big.sum_logprob.cpu().numpy().copy()
    for lo in (0, 16, 32):
        hi = min(37, lo + 16)
        part = eng.decode(enc[lo:hi].contiguous(), tok, tok.sot_sequence(), **kw)
        assert np.array_equal(part.tokens.cpu().numpy()[: hi - lo], tb[lo:hi]), lo
        # bit-identical log-probabilities too: the final LayerNorm of a wide launch uses the fused prologue's arithmetic
        assert np.array_equal(part.sum_logprob.cpu().numpy()[: hi - lo], lb[lo:hi]), lo
    # the balanced kernels at <= 16 rows (variant 3) against the default step (variant 1)
    v3 = eng.decode(enc[:9].contiguous(), tok, tok.sot_sequence(), step_variant=3, **kw)
    assert np.array_equal(v3.tokens.cpu().numpy()[:9], tb[:9])
    # and against the oracle on a few rows
    part = eng.decode(enc[33:37].contiguous(), tok, tok.sot_sequence(), **kw)
    assert np.array_equal(part.tokens.cpu().numpy()[:4], tb[33:37])
    _strict(ck, DIMS, enc[33:37], part, tok, OD.RULES_LIGHTNING, tok.suppress_tokens())


def test_greedy_decode_up_to_64_rows():
    """the largest launch the context takes (4 row groups, the last one ragged)"""
    eng, _ = G.tiny_engine(max_batch=64)
    tok = get_tokenizer(DIMS.n_vocab)
    mel = _mel(53, seed=12)
    enc = eng.encode(mel.cuda())
    kw = dict(rules=E.RULES_LIGHTNING, suppress_ids=tok.suppress_tokens(), sample_len=24, check_every=8)
    big = eng.decode(enc, tok, tok.sot_sequence(), **kw)
    tb, lb = big.tokens.cpu().numpy().copy(), big.sum_logprob.cpu().numpy().copy()
    eng.check_status()
    for lo in (0, 48):
        hi = min(53, lo + 16)
        part = eng.decode(enc[lo:hi].contiguous(), tok, tok.sot_sequence(), **kw)
        assert np.array_equal(part.tokens.cpu().numpy()[: hi - lo], tb[lo:hi]), lo
        assert np.array_equal(part.sum_logprob.cpu().numpy()[: hi - lo], lb[lo:hi]), lo


def test_greedy_decode_up_to_128_rows():
    """the largest launch a context takes since the scheduler cuts 128-row passes: eight row groups (the last one ragged),
    the logits GEMV as two launches of <= 64 rows -- every row the tokens and the log-probability of a 16-row launch"""
    eng, _ = G.tiny_engine(max_batch=128)
    tok = get_tokenizer(DIMS.n_vocab)
    mel = _mel(117, seed=13)
    enc = eng.encode(mel.cuda())
    kw = dict(rules=E.RULES_LIGHTNING, suppress_ids=tok.suppress_tokens(), sample_len=24, check_every=8)
    big = eng.decode(enc, tok, tok.sot_sequence(), **kw)
    tb, lb = big.tokens.cpu().numpy().copy(), big.sum_logprob.cpu().numpy().copy()
    eng.check_status()
    for lo in (0, 64, 112):
        hi = min(117, lo + 16)
        part = eng.decode(enc[lo:hi].contiguous(), tok, tok.sot_sequence(), **kw)
        assert np.array_equal(part.tokens.cpu().numpy()[: hi - lo], tb[lo:hi]), lo
        assert np.array_equal(part.sum_logprob.cpu().numpy()[: hi - lo], lb[lo:hi]), lo
    v1 = eng.decode(enc, tok, tok.sot_sequence(), step_variant=1, **kw)         # one kernel per stage at 128 rows too
    assert np.array_equal(v1.tokens.cpu().numpy(), tb) and np.array_equal(v1.sum_logprob.cpu().numpy(), lb)
    # the launch shape of a pass with other passes in flight (fc2_tile_n = 16): from 64 rows on the decode GEMVs run as one
    # pass over the weights for four row groups (skinny_wide_kernel) -- the same bits, here and at exactly 64 rows
    for rows in (117, 64):
        sub = enc[:rows].contiguous()
        if rows == 117:
            rt, rl = tb, lb
        else:                                   # (decode() hands out views of the engine's buffers: copy before the next call)
            ref = eng.decode(sub, tok, tok.sot_sequence(), **kw)
            rt, rl = ref.tokens.cpu().numpy().copy(), ref.sum_logprob.cpu().numpy().copy()
        op = eng.decode(sub, tok, tok.sot_sequence(), fc2_tile_n=16, **kw)
        assert np.array_equal(op.tokens.cpu().numpy()[:rows], rt[:rows]), rows
        assert np.array_equal(op.sum_logprob.cpu().numpy()[:rows], rl[:rows]), rows


def test_greedy_decode_int8_decoder_weights():
    """compute_type int8: decoder GEMV weights as int8 + row scales (last layer kept fp16), against the
    oracle's fp32 decoder on the quantise -> dequantise checkpoint."""
    from oracle import quant as OQ
    from whisperx_mlx_amd import weights as WT
    ck = WT.random_checkpoint(DIMS, seed=0, std=0.2, emb_std=0.1)
    packed = WT.quantize_packed_decoder(WT.pack(ck, DIMS, "cuda"), DIMS)
    assert "dec.0.qkv.wq" in packed and "dec.0.qkv.w" not in packed and f"dec.{DIMS.n_text_layer - 1}.qkv.w" in packed
    eng = E.WhisperHipEngine(DIMS, packed, max_batch=4, alignment_heads=G.TEST_HEADS)
    ckq = OQ.dequantized_checkpoint({k: v.float() for k, v in ck.items()}, DIMS.n_text_layer)
    tok = get_tokenizer(DIMS.n_vocab)
    sp = OD.Specials.for_vocab(DIMS.n_vocab)
    mel = _mel(4, seed=21)
    enc = eng.encode(mel.cuda())
    out = eng.decode(enc, tok, tok.sot_sequence(), rules=E.RULES_LIGHTNING, suppress_ids=tok.suppress_tokens(), sample_len=32)
    eng.check_status()
    _strict(ckq, DIMS, enc, out, tok, OD.RULES_LIGHTNING, tok.suppress_tokens())
    # and it is a different model from the fp16 one: quantisation must have moved the logits
    lg16 = G.tiny_engine()[0].decode_logits(enc, torch.tensor([list(tok.sot_sequence())] * 4, dtype=torch.int32).cuda())
    lg8 = eng.decode_logits(enc, torch.tensor([list(tok.sot_sequence())] * 4, dtype=torch.int32).cuda())
    assert float((lg16 - lg8).abs().max()) > 1e-4


def test_greedy_forced_len_and_determinism():
    eng, ck = tiny()
    tok = get_tokenizer(DIMS.n_vocab)
    mel = _mel(4, seed=4)
    enc = eng.encode(mel.cuda())
    a = eng.decode(enc, tok, tok.sot_sequence(), rules=0, forced_len=37, use_graph=True)
    ta = a.tokens.cpu().numpy().copy()
    assert a.n_sampled == 37
    b = eng.decode(enc, tok, tok.sot_sequence(), rules=0, forced_len=37, use_graph=False)
    tb = b.tokens.cpu().numpy().copy()
    assert np.array_equal(ta, tb)                       # graph replay == direct launches, run to run
    c = eng.decode(enc, tok, tok.sot_sequence(), rules=0, forced_len=37, use_graph=True, fc2_tile_n=16, cross_split=4)
    assert np.array_equal(ta, c.tokens.cpu().numpy())   # block shapes (FC2 tile, key splits) do not change a token
    P = len(tok.sot_sequence())
    assert (ta[:, P: P + 37] != tok.eot).all() and (ta[:, P + 37:] == tok.eot).all()


@pytest.mark.parametrize("mode", [0, 1])
def test_cross_attention_capture_and_dtw(mode):
    eng, ck = tiny()
    tok = get_tokenizer(DIMS.n_vocab)
    sp = OD.Specials.for_vocab(DIMS.n_vocab)
    mel = _mel(2, seed=6)
    enc = eng.encode(mel.cuda())
    out = eng.decode(enc, tok, tok.sot_sequence(), rules=E.RULES_LIGHTNING, suppress_ids=tok.suppress_tokens(),
                     sample_len=24, capture_qk=True, use_graph=True)
    qk = eng.align_qk(2).cpu().numpy()                  # (B, heads, 224, 1500)
    toks = out.tokens.cpu().numpy()
    P = out.n_prompt
    # oracle scores for the SAME token sequence (teacher forced), alignment heads only
    xkv = OW.cross_kv(ck, DIMS, enc.float().cpu())
    for b in range(2):
        seq = torch.from_numpy(toks[b: b + 1, : P + 24].astype(np.int64))
        _, _, cq = OW.decoder_forward(ck, DIMS, seq, [(k[b: b + 1], v[b: b + 1]) for k, v in xkv])
        for hi, (l, h) in enumerate(G.TEST_HEADS):
            ref = cq[l][0, h, P - 1: P - 1 + 24].numpy()      # query position P-1+s produced sampled token s
            # fp16 q/k projections vs fp32: relative to the score magnitude (|qk| ~ 10)
            assert np.abs(qk[b, hi, :24] - ref).max() < 5e-3 * np.abs(ref).max() + 1e-2
    paths, mat = eng.dtw_path(out, tok.eot, mode=mode, want_matrix=True)
    mat = mat.cpu().numpy()
    for b in range(2):
        sampled = toks[b, P: P + 24].tolist()
        rows = []
        for s, t in enumerate(sampled):
            if t < tok.eot:
                rows.append(s)
            elif t == tok.eot:
                rows.append(s)
                break
        n_rows, path = paths[b]
        assert n_rows == len(rows)
        if n_rows < 2:
            continue
        sel = qk[b][:, rows, :]
        if mode == 0:
            ref_m = ODTW.alignment_matrix_upstream(sel)
            assert np.abs(mat[b, :n_rows] - ref_m).max() < 5e-3
            ref_path = ODTW.dtw_path_fast(-mat[b, :n_rows])     # DP on the GPU's own matrix: integer parity
        else:
            ref_m = ODTW.alignment_matrix_inrepo(sel)
            assert np.abs(mat[b, :n_rows] - ref_m).max() < 5e-3
            ref_path = ODTW.dtw_path_fast(-mat[b, :n_rows].T)
        assert np.array_equal(path, ref_path)


def test_hip_engine_on_transformers_checkpoint_dir(tmp_path):
    """A transformers checkpoint directory loaded by the product's loader, run by the HIP engine, against the
    transformers model's own encoder output and logits (third-party implementation, fp32 on the CPU)."""
    import warnings
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    from whisperx_mlx_amd import weights as WT
    warnings.filterwarnings("ignore")
    torch.manual_seed(1)
    cfg = WhisperConfig(vocab_size=600, num_mel_bins=80, encoder_layers=2, decoder_layers=3, encoder_attention_heads=2,
                        decoder_attention_heads=2, d_model=128, encoder_ffn_dim=512, decoder_ffn_dim=512,
                        max_source_positions=1500, max_target_positions=448, activation_function="gelu",
                        pad_token_id=0, bos_token_id=1, eos_token_id=2, decoder_start_token_id=3)
    m = WhisperForConditionalGeneration(cfg).eval().half().float()      # weights exactly representable in fp16
    m.save_pretrained(tmp_path, safe_serialization=True)
    dims, sd, _ = WT.load_checkpoint_dir(str(tmp_path))
    eng = E.WhisperHipEngine(dims, WT.pack(sd, dims, "cuda"), max_batch=3)
    mel = (torch.randn(3, 3000, 80) * 0.5).half()
    toks = torch.randint(0, 600, (3, 7), dtype=torch.int32)
    with torch.no_grad():
        enc_hf = m.model.encoder(mel.float().permute(0, 2, 1)).last_hidden_state
        lg_hf = m(input_features=mel.float().permute(0, 2, 1), decoder_input_ids=toks.long()).logits[:, -1]
    enc = eng.encode(mel.cuda())
    assert G.rel_err(enc, enc_hf.cuda()) < ENC_TOL
    lg = eng.decode_logits(enc, toks.cuda())
    eng.check_status()
    assert float((lg.cpu() - lg_hf).abs().max()) < 3e-2 * max(1.0, float(lg_hf.abs().max()))
    assert (lg.cpu().argmax(-1) == lg_hf.argmax(-1)).float().mean() >= 2 / 3


def test_mid_size_model_takes_the_256_tile_gemm():
    """d = 512 (8 heads), 2 + 2 layers: every encoder GEMM, the conv2 implicit GEMM and the head-split cross-K/V store run
    on the 256 x 256 kernel (the 2-layer d = 128 test model never reaches it); encoder output and teacher-forced logits
    against the oracle."""
    from whisperx_mlx_amd import weights as WT
    dims = WT.ModelDimensions(80, 1500, 512, 8, 2, 51865, 448, 512, 8, 2)
    ck = WT.random_checkpoint(dims, seed=4, std=0.05, emb_std=0.05)
    eng = E.WhisperHipEngine(dims, WT.pack(ck, dims, "cuda"), max_batch=3)
    ck32 = {k: v.float() for k, v in ck.items()}
    g = torch.Generator().manual_seed(9)
    mel = (torch.randn(3, 3000, 80, generator=g) * 0.5).half()
    enc = eng.encode(mel.cuda())
    ref = OW.encoder_forward(ck32, dims, mel.float())
    assert G.rel_err(enc, ref) < ENC_TOL
    tok = get_tokenizer(dims.n_vocab)
    toks = torch.tensor([list(tok.sot_sequence()) + [1000 + i, 2000, 3000 + i] for i in range(3)], dtype=torch.int32)
    lg = eng.decode_logits(enc, toks.cuda()).cpu()
    eng.check_status()
    lg_ref, _, _ = OW.decoder_forward(ck32, dims, toks.long(), OW.cross_kv(ck32, dims, enc.float().cpu()))
    lg_ref = lg_ref[:, -1]
    assert float((lg - lg_ref).abs().max()) < 3e-2 * max(1.0, float(lg_ref.abs().max()))


def test_greedy_decode_long_sequences_vs_oracle():
    """150 sampled tokens: the self-attention single-trip path for more than 64 cached keys (and the key-split merge over
    many positions) against the oracle's step-by-step decode."""
    eng, ck = tiny()
    tok = get_tokenizer(DIMS.n_vocab)
    sp = OD.Specials.for_vocab(DIMS.n_vocab)
    mel = _mel(2, seed=31)
    enc = eng.encode(mel.cuda())
    out = eng.decode(enc, tok, tok.sot_sequence(), rules=0, forced_len=150, cross_split=2)
    eng.check_status()
    assert out.n_sampled == 150
    rep = _strict(ck, DIMS, enc, out, tok, 0, forced_len=150)
    assert rep.steps_checked == 2 * 150


@pytest.mark.parametrize("rows", [5, 16, 37])
def test_fused_launches_equal_one_kernel_per_stage(rows):
    """step variant 0 / 4 (dependent stages of a layer share a launch and hand over through tagged granules,
    csrc/declayer.hip) against variant 1 (one kernel per stage): the arithmetic and its order are the same, so tokens,
    log-probabilities and the captured alignment-head scores must be bit-identical -- at 5 rows, a full tile of 16 and
    three row groups (37)."""
    eng, ck = G.tiny_engine(max_batch=40)
    tok = get_tokenizer(DIMS.n_vocab)
    enc = eng.encode(_mel(rows, seed=41).cuda())
    kw = dict(rules=E.RULES_LIGHTNING, suppress_ids=tok.suppress_tokens(), sample_len=36, capture_qk=True, cross_split=2)
    a = eng.decode(enc, tok, tok.sot_sequence(), step_variant=1, **kw)
    ta, la, qa = a.tokens.cpu().numpy().copy(), a.sum_logprob.cpu().numpy().copy(), eng.align_qk(rows).cpu().numpy().copy()
    eng.check_status()
    for use_graph in (True, False):
        b = eng.decode(enc, tok, tok.sot_sequence(), step_variant=0, use_graph=use_graph, **kw)
        eng.check_status()
        assert a.n_sampled == b.n_sampled
        assert np.array_equal(b.tokens.cpu().numpy(), ta) and np.array_equal(b.sum_logprob.cpu().numpy(), la)
        assert np.array_equal(eng.align_qk(rows).cpu().numpy(), qa)
    if rows <= 16:
        _strict(ck, DIMS, enc, b, tok, OD.RULES_LIGHTNING, tok.suppress_tokens())


def test_sampler_ends_a_poisoned_row_with_eot():
    """a row whose logits are all NaN (what a bounded in-kernel wait that gave up leaves behind; the device flag reports
    it) must end with EOT: an out-of-range id must never reach the next step's embedding lookup"""
    eng, _ = tiny()
    tok = get_tokenizer(DIMS.n_vocab)
    mask = eng.suppress_mask(sorted(set(tok.suppress_tokens()) | {tok.no_timestamps}))
    o = _opts(tok, OD.RULES_LIGHTNING, tok.suppress_tokens(), tok.sot_sequence(), mask)
    P = len(tok.sot_sequence())
    tokens = torch.full((3, DIMS.n_text_ctx), tok.eot, dtype=torch.int32)
    tokens[:, :P] = torch.tensor(tok.sot_sequence(), dtype=torch.int32)
    tokens = tokens.cuda()
    lg = torch.zeros(3, DIMS.n_vocab, device="cuda")
    lg[1] = float("nan")
    lg[0, tok.timestamp_begin + 3] = 9.0
    lg[2, tok.timestamp_begin + 7] = 9.0
    slp = torch.zeros(3, device="cuda")
    nsp = torch.zeros(3, device="cuda")
    rc = _lib.lib().wx_sample_step(eng.ctx, _lib.ptr(lg), lg.stride(0), _lib.ptr(tokens), DIMS.n_text_ctx, P, 3, C.byref(o),
                                   _lib.ptr(slp), _lib.ptr(nsp), None)
    _lib.check(eng.ctx, rc, "wx_sample_step")
    torch.cuda.synchronize()
    assert tokens[:, P].cpu().tolist() == [tok.timestamp_begin + 3, tok.eot, tok.timestamp_begin + 7]


def test_finished_rows_sit_out_without_changing_the_others():
    """rows that have emitted EOT take no part in the attention kernels any more (the reference forwards active sequences
    only, mlx_whisper_batch_decoder.py:361-373).  With per-row forced lengths (bench workload option forced_lens) every row
    must end exactly where told, and until then decode exactly the tokens it decodes when all rows run to the end."""
    eng, ck = tiny()
    tok = get_tokenizer(DIMS.n_vocab)
    enc = eng.encode(_mel(4, seed=51).cuda())
    kw = dict(rules=E.RULES_LIGHTNING, suppress_ids=tok.suppress_tokens(), forced_len=60, capture_qk=True)
    full = eng.decode(enc, tok, tok.sot_sequence(), **kw)
    tf, qf = full.tokens.cpu().numpy().copy(), eng.align_qk(4).cpu().numpy().copy()
    _strict(ck, DIMS, enc, full, tok, OD.RULES_LIGHTNING, tok.suppress_tokens(), forced_len=60)     # (views the engine's buffers: before the next decode)
    lens = [60, 7, 33, 1]
    part = eng.decode(enc, tok, tok.sot_sequence(), forced_lens=torch.tensor(lens, dtype=torch.int32).cuda(), **kw)
    eng.check_status()
    tp, qp = part.tokens.cpu().numpy(), eng.align_qk(4).cpu().numpy()
    P = full.n_prompt
    for b, n in enumerate(lens):
        assert np.array_equal(tp[b, : P + n], tf[b, : P + n]), b            # same tokens while the row is alive
        assert (tp[b, P + n:] == tok.eot).all(), b                             # then EOT for good
        assert np.array_equal(qp[b, :, :n], qf[b, :, :n]), b                  # captured scores of the live steps unchanged


def test_english_only_vocabulary_greedy_strict():
    """*.en models: gpt2 vocabulary (51864 entries), no language tokens, a one-token prompt [sot], EOT = 50256 -- the
    prompt / first-sampled-step bookkeeping differs from the multilingual models (n_prompt = 1)"""
    from whisperx_mlx_amd import weights as WT
    dims = WT.ModelDimensions(80, 1500, 128, 2, 2, 51864, 448, 128, 2, 2)
    ck = WT.random_checkpoint(dims, seed=5, std=0.2, emb_std=0.1)
    eng = E.WhisperHipEngine(dims, WT.pack(ck, dims, "cuda"), max_batch=4, alignment_heads=G.TEST_HEADS)
    tok = get_tokenizer(dims.n_vocab)
    assert tok.sot_sequence() == [50257] and tok.eot == 50256 and not tok.is_multilingual and tok.timestamp_begin == 50363
    sp = OD.Specials(n_vocab=51864, eot=tok.eot, sot=tok.sot, n_langs=0, translate=tok.translate, transcribe=tok.transcribe,
                     sot_lm=tok.sot_lm, sot_prev=tok.sot_prev, no_speech=tok.no_speech, no_timestamps=tok.no_timestamps,
                     timestamp_begin=tok.timestamp_begin)
    enc = eng.encode(_mel(4, seed=61).cuda())
    out = eng.decode(enc, tok, tok.sot_sequence(), rules=E.RULES_LIGHTNING, suppress_ids=tok.suppress_tokens(), sample_len=30,
                     capture_qk=True)
    eng.check_status()
    assert out.n_prompt == 1
    ck32 = {k: v.float() for k, v in ck.items()}
    rep = PAR.check_tokens_strict(ck32, dims, enc, out.tokens.cpu().numpy(), 1, out.n_sampled, sp, OD.RULES_LIGHTNING,
                                  tok.suppress_tokens(), tol=MARGIN_TOL, gpu_sum_logprob=out.sum_logprob.cpu().numpy(), lp_tol=0.01)
    PAR.assert_strict(rep)
    first = out.tokens.cpu().numpy()[:, 1]
    assert ((first >= tok.timestamp_begin) & (first <= tok.timestamp_begin + 50)).all()      # initial-timestamp rule
    paths = eng.dtw_path(out, tok.eot)
    assert len(paths) == 4


def test_padding_rows_change_nothing_for_the_chunks():
    """One launch shape per context: a pass of n chunks launched with `rows` > n rows (wx_decode_opts.n_active: the extra
    rows are padding that counts as finished, needs no encoder output and sits out of the attention kernels) gives the
    chunks exactly the tokens, log-probabilities, no-speech probabilities, step count and captured alignment scores of
    an n-row launch -- forced lengths and free-running (EOT-terminated) alike, whatever the cross-K/V cache held."""
    eng, ck = G.tiny_engine(max_batch=8)
    tok = get_tokenizer(DIMS.n_vocab)
    enc8 = eng.encode(_mel(8, seed=71).cuda())
    eng.decode(enc8, tok, tok.sot_sequence(), rules=0, forced_len=20)            # leaves all 8 rows of the caches dirty
    for n in (5, 1):
        enc = enc8[:n].contiguous()
        for kw in (dict(rules=E.RULES_LIGHTNING, suppress_ids=tok.suppress_tokens(), forced_len=40, capture_qk=True),
                   dict(rules=E.RULES_LIGHTNING, suppress_ids=tok.suppress_tokens(), sample_len=48, capture_qk=True, check_every=4),
                   dict(rules=0, sample_len=40, check_every=8, step_variant=1)):
            a = eng.decode(enc, tok, tok.sot_sequence(), **kw)
            eng.check_status()
            ta, la, na, sa = a.tokens.cpu().numpy().copy(), a.sum_logprob.cpu().numpy().copy(), a.no_speech_prob.cpu().numpy().copy(), a.n_sampled
            qa = eng.align_qk(n).cpu().numpy().copy() if kw.get("capture_qk") else None
            b = eng.decode(enc, tok, tok.sot_sequence(), rows=8, **kw)
            eng.check_status()
            assert b.tokens.shape[0] == n and b.n_sampled == sa, (n, kw)
            assert np.array_equal(b.tokens.cpu().numpy(), ta), (n, kw)
            assert np.array_equal(b.sum_logprob.cpu().numpy(), la) and np.array_equal(b.no_speech_prob.cpu().numpy(), na), (n, kw)
            if qa is not None:
                assert np.array_equal(eng.align_qk(n).cpu().numpy(), qa), (n, kw)
    # and the padded launch is what the oracle decodes
    out = eng.decode(enc8[:3].contiguous(), tok, tok.sot_sequence(), rules=E.RULES_LIGHTNING, suppress_ids=tok.suppress_tokens(),
                     sample_len=32, rows=8)
    _strict(ck, DIMS, enc8[:3].contiguous(), out, tok, OD.RULES_LIGHTNING, tok.suppress_tokens())
