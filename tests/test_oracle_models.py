"""CPU: the parts of the oracle that cannot be pinned by reference fixtures (the arithmetic
lives in third-party mlx-whisper, see oracle/__init__.py) are cross-checked against
HuggingFace transformers -- same published architecture / algorithms, seeded random weights."""
import warnings

import numpy as np
import pytest
import torch

from oracle import decoding as OD
from oracle import dtw as ODTW
from oracle import wav2vec2_ref as OWV
from oracle import whisper_ref as OW

warnings.filterwarnings("ignore")


@pytest.fixture(scope="module")
def hf_whisper():
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    torch.manual_seed(0)
    cfg = WhisperConfig(vocab_size=600, num_mel_bins=16, encoder_layers=2, decoder_layers=2, encoder_attention_heads=2,
                        decoder_attention_heads=2, d_model=128, encoder_ffn_dim=512, decoder_ffn_dim=512,
                        max_source_positions=1500, max_target_positions=448, activation_function="gelu",
                        pad_token_id=0, bos_token_id=1, eos_token_id=2, decoder_start_token_id=3)
    m = WhisperForConditionalGeneration(cfg).eval()
    dims = OW.Dims(16, 1500, 128, 2, 2, 600, 448, 128, 2, 2)
    return m, dims, OW.from_hf_state_dict(m.state_dict(), dims)


def test_whisper_encoder_decoder_vs_transformers(hf_whisper):
    m, dims, w = hf_whisper
    mel = torch.randn(2, 3000, 16)
    toks = torch.randint(0, 600, (2, 6))
    with torch.no_grad():
        enc_hf = m.model.encoder(mel.permute(0, 2, 1)).last_hidden_state
        out = m(input_features=mel.permute(0, 2, 1), decoder_input_ids=toks)
    enc = OW.encoder_forward(w, dims, mel)
    assert (enc - enc_hf).abs().max() < 1e-5
    xkv = OW.cross_kv(w, dims, enc)
    lg, cache, qk = OW.decoder_forward(w, dims, toks, xkv)
    assert (lg - out.logits).abs().max() < 1e-5
    # incremental decoding with the KV cache == full forward
    lg4, cache4, _ = OW.decoder_forward(w, dims, toks[:, :4], xkv)
    lg5, cache5, _ = OW.decoder_forward(w, dims, toks[:, 4:5], xkv, cache4, 4)
    lg6, _, qk6 = OW.decoder_forward(w, dims, toks[:, 5:6], xkv, cache5, 5)
    assert (lg6[:, 0] - lg[:, 5]).abs().max() < 1e-5
    assert (qk6[1][:, :, 0] - qk[1][:, :, 5]).abs().max() < 1e-5


def test_wav2vec2_vs_transformers():
    from transformers import Wav2Vec2Config, Wav2Vec2ForCTC
    torch.manual_seed(0)
    cfg = Wav2Vec2Config(vocab_size=32, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                         conv_dim=(64,) * 7, num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16,
                         feat_extract_norm="group", do_stable_layer_norm=False, conv_bias=False)
    m = Wav2Vec2ForCTC(cfg).eval()
    sd = OWV.fold_weight_norm({k: v.float() for k, v in m.state_dict().items()})
    dims = OWV.W2VDims(conv_dim=64, hidden=128, heads=2, layers=2, ffn=256, vocab=32)
    for n in (16000, 4321, 400):
        x = torch.randn(1, n) * 0.1
        with torch.no_grad():
            ref = m(x).logits
        out = OWV.forward_logits(sd, dims, x)
        assert out.shape == ref.shape and out.shape[1] == OWV.n_frames(n, dims)
        assert (out - ref).abs().max() < 1e-5


def test_wav2vec2_layernorm_stable_variant_vs_transformers():
    """the large / XLSR family most non-English align models use (alignment.py:39-74)"""
    from transformers import Wav2Vec2Config, Wav2Vec2ForCTC
    torch.manual_seed(1)
    cfg = Wav2Vec2Config(vocab_size=40, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                         conv_dim=(64,) * 7, num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16,
                         feat_extract_norm="layer", do_stable_layer_norm=True, conv_bias=True)
    m = Wav2Vec2ForCTC(cfg).eval()
    sd = OWV.fold_weight_norm({k: v.float() for k, v in m.state_dict().items()})
    dims = OWV.W2VDims(conv_dim=64, hidden=128, heads=2, layers=2, ffn=256, vocab=40, norm_mode=1, stable_ln=1)
    x = torch.randn(1, 9000) * 0.1
    with torch.no_grad():
        ref = m(x).logits
    assert (OWV.forward_logits(sd, dims, x) - ref).abs().max() < 1e-5


def test_timestamp_rules_vs_transformers_processor():
    """ApplyTimestampRules restatement == HF WhisperTimeStampLogitsProcessor on random histories
    (HF folds the same published rules; sample_begin / max_initial_timestamp handled explicitly)."""
    from transformers.generation.logits_process import WhisperTimeStampLogitsProcessor
    sp = OD.Specials.for_vocab(51865)
    gen_cfg = type("G", (), dict(eos_token_id=sp.eot, no_timestamps_token_id=sp.no_timestamps,
                                 max_initial_timestamp_index=50, forced_decoder_ids=None))()
    begin = 3
    proc = WhisperTimeStampLogitsProcessor(gen_cfg, begin_index=begin)
    g = torch.Generator().manual_seed(1)
    tb = sp.timestamp_begin
    histories = [[], [tb + 5], [tb + 5, 100], [tb + 5, 100, 200, tb + 40], [tb + 5, 100, tb + 40, tb + 40],
                 [tb + 5, 100, tb + 40, tb + 40, 7, 8], [tb, tb], [tb + 10, 99, 98, 97, tb + 1499]]
    rules = OD.RULE_TS_NOTIMESTAMPS | OD.RULE_TS_PAIRS | OD.RULE_TS_MONOTONE | OD.RULE_TS_INITIAL | OD.RULE_TS_PROB
    for h in histories:
        toks = torch.tensor([[sp.sot, sp.lang_token(0), sp.transcribe] + h])
        logits = torch.randn(1, 51865, generator=g) * 3
        ref = proc(toks, logits.clone())
        got = OD.apply_filters(logits.clone(), toks, sp, begin, rules, max_initial_timestamp_index=50)
        assert torch.equal(torch.isinf(ref), torch.isinf(got)), h
        assert int(ref.argmax()) == int(got.argmax()), h


def test_dtw_and_median_vs_transformers():
    from transformers.models.whisper.generation_whisper import _dynamic_time_warping, _median_filter
    rng = np.random.default_rng(0)
    for shape in ((5, 40), (30, 200), (1, 17), (12, 12)):
        x = rng.normal(size=shape).astype(np.float32)
        ti, fi = _dynamic_time_warping(-x.astype(np.float64))
        p = ODTW.dtw_path(-x)
        p2 = ODTW.dtw_path_fast(-x)
        assert np.array_equal(p[0], ti) and np.array_equal(p[1], fi)
        assert np.array_equal(p, p2)
    w = torch.from_numpy(rng.normal(size=(3, 7, 50)).astype(np.float32))
    ref = _median_filter(w, 7).numpy()
    assert np.allclose(ODTW.median_filter_reflect(w.numpy(), 7), ref)


def test_greedy_update_semantics():
    """mlx_whisper_batch_decoder.py:267-303: EOT pinning and logprob accumulation."""
    eot = 9
    tokens = torch.tensor([[1, 2, 3], [1, 2, eot]])
    logits = torch.tensor([[0.0, 2.0, 1.0, 0, 0, 0, 0, 0, 0, 0.5], [5.0, 0, 0, 0, 0, 0, 0, 0, 0, 0.0]])
    t2, done, slp = OD.greedy_update(tokens, logits, torch.zeros(2), eot)
    assert t2[:, -1].tolist() == [1, eot] and done.tolist() == [False, True]
    lp = logits[0] - torch.logsumexp(logits[0], 0)
    assert abs(float(slp[0]) - float(lp[1])) < 1e-6 and float(slp[1]) == 0.0
