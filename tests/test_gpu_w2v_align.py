"""-m gpu: wav2vec2 CTC forward (padded ragged batch) against the oracle, and the full
align() on the GPU against align() driven by the oracle on the same weights."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ctc as OC                   # noqa: E402
from oracle import wav2vec2_ref as OWV         # noqa: E402
from tests.conftest import GOLDEN              # noqa: E402
from tests.synth import speechlike_audio, synth_audio            # noqa: E402
from whisperx_mlx_amd import alignment as AL   # noqa: E402
from whisperx_mlx_amd.w2v import W2VConfig, W2VHipModel   # noqa: E402

CFG = W2VConfig(conv_dim=64, hidden=128, heads=2, layers=2, ffn=256, vocab=32)
ODIMS = OWV.W2VDims(conv_dim=64, hidden=128, heads=2, layers=2, ffn=256, vocab=32)
EMIS_TOL = 3e-2      # abs on log-probs (fp16 activations through 7 convs + 2 layers vs fp32)


def _model():
    if not hasattr(_model, "m"):
        w = OWV.random_weights(ODIMS, seed=3)
        _model.m = (W2VHipModel.from_state_dict(w, CFG), w)
    return _model.m


def test_w2v_frames_formula():
    for n in (400, 401, 16000, 47999, 480000):
        assert CFG.n_frames(n) == OWV.n_frames(n, ODIMS)


def test_w2v_emissions_ragged_batch_vs_oracle():
    m, w = _model()
    waves = [synth_audio(5, 16000), synth_audio(6, 4000), synth_audio(7, 33333), synth_audio(8, 300), synth_audio(9, 16000)]
    logp, T = m.emissions(waves)
    torch.cuda.synchronize()
    logp = logp.cpu()
    for i, wv in enumerate(waves):
        ref = OWV.emissions(w, ODIMS, torch.from_numpy(wv))
        assert T[i] == ref.shape[0]
        got = logp[i, : T[i]]
        assert torch.isfinite(got).all()
        assert (got - ref).abs().max().item() < EMIS_TOL, i
        # a segment's result must not depend on its batch mates: run it alone
        if i in (1, 3):
            solo, Ts = m.emissions([wv])
            assert Ts[0] == T[i]
            assert (solo[0, : T[i]].cpu() - got).abs().max().item() < 2e-3


def test_w2v_emissions_are_run_to_run_bit_identical():
    """the GroupNorm statistics of conv0 are reduced in a fixed order (round 4: signal sums in double, no atomics; the first
    version summed per-block partials with float64 atomics, whose order the hardware picks): the same batch twice gives the
    same bits, also when another batch ran in between"""
    m, _w = _model()
    waves = [synth_audio(15, 48000), synth_audio(16, 4100), synth_audio(17, 33333)]
    a, Ta = m.emissions(waves)
    a = a.clone()
    m.emissions([synth_audio(18, 20000)] * 4)
    b, Tb = m.emissions(waves)
    assert Ta == Tb
    for i, t in enumerate(Ta):                    # (frames past a segment's own length are padding: unspecified)
        assert torch.equal(a[i, :t], b[i, :t]), i


def test_w2v_batch_capacity_then_smaller_batch():
    """buffers are sized by the largest batch seen; a later smaller batch must still be right"""
    m, w = _model()
    m.emissions([synth_audio(1, 50000)] * 3)
    wv = synth_audio(2, 9000)
    logp, T = m.emissions([wv])
    ref = OWV.emissions(w, ODIMS, torch.from_numpy(wv))
    assert (logp[0, : T[0]].cpu() - ref).abs().max().item() < EMIS_TOL


def test_align_end_to_end_gpu_vs_oracle_driven():
    """same transcript, same weights: align() with the HIP model vs align() whose emissions
    and DP come from the oracle.  Word boundaries must agree within +-20 ms."""
    # a sharpened CTC head: plain random weights give nearly flat emissions, whose DP decisions are near-ties that fp16
    # activations move; with the head scaled the frames commit to labels and EVERY timed word must agree within +-20 ms
    # (tests/test_gpu_at_size.py runs the same comparison at wav2vec2-base size over a few hundred words)
    w = dict(OWV.random_weights(ODIMS, seed=3))
    w["lm_head.weight"] = (w["lm_head.weight"] * 8.0).half().float()
    m = W2VHipModel.from_state_dict(w, CFG)
    audio = np.load(os.path.join(GOLDEN, "logmel.npz"))["audio_sample_i16"].astype(np.float32) / 32768.0
    labels = ["<pad>", "<s>", "</s>", "<unk>", "|"] + list("etaonihsrdlumwcfgypbvk'xjqz")
    meta = {"language": "en", "dictionary": {c.lower(): i for i, c in enumerate(labels)}, "type": "hip"}
    segs = [{"start": 0.976, "end": 2.539, "text": "That's why he's so fucking famous, bro."},
            {"start": 3.681, "end": 5.0, "text": "That's why Gordon Ramsey's so famous."},
            {"start": 0.2, "end": 0.9, "text": " 3 ok. Fine! "},
            {"start": 6.0, "end": 7.0, "text": "beyond"}]

    def oracle_aligner(waveforms, token_lists, blank_id, beam):
        out = []
        for wav, toks in zip(waveforms, token_lists):
            em = OWV.emissions(w, ODIMS, torch.from_numpy(np.asarray(wav))).numpy()
            tr = OC.get_trellis(em, toks, blank_id)
            path = OC.backtrack_beam(tr, em, toks, blank_id, beam)
            out.append((em.shape[0], None, None) if path is None else
                       (em.shape[0], [p[0] for p in path], [p[2] for p in path]))
        return out

    got = AL.align([dict(s) for s in segs], m, meta, audio, "cuda")
    ref = AL.align([dict(s) for s in segs], None, meta, audio, "cpu", _aligner=oracle_aligner)
    assert len(got["segments"]) == len(ref["segments"])
    assert [w_["word"] for w_ in got["word_segments"]] == [w_["word"] for w_ in ref["word_segments"]]
    n_close = 0
    for a, b in zip(got["word_segments"], ref["word_segments"]):
        assert ("start" in a) == ("start" in b)
        if "start" in a:
            n_close += abs(a["start"] - b["start"]) <= 0.0201 and abs(a["end"] - b["end"]) <= 0.0201
    n_timed = sum("start" in b for b in ref["word_segments"])
    assert n_timed >= 10 and n_close == n_timed, (n_close, n_timed)


def test_align_gpu_on_reference_emissions_exact():
    """the GPU CTC kernels inside align(): with the reference's own emissions (fixture) the
    whole result dict must equal the reference's result exactly."""
    m, _ = _model()
    with open(os.path.join(GOLDEN, "align.json")) as f:
        docs = json.load(f)
    for name in ("short_json", "edge_cases"):
        doc = docs[name]
        em = np.load(os.path.join(GOLDEN, f"align_{name}_emissions.npz"))
        emissions = [torch.from_numpy(em[f"call{i}"]) for i in range(len(em.files))]
        calls = iter(emissions)

        def gpu_aligner(waveforms, token_lists, blank_id, beam):
            out = []
            for toks in token_lists:
                e = next(calls)
                ptok, pscore, ok, _ = m.ctc_align(e[None], torch.tensor([e.shape[0]]), torch.tensor([toks], dtype=torch.int32),
                                                  torch.tensor([len(toks)]), blank_id, beam)
                T = e.shape[0]
                out.append((T, ptok[0, :T].cpu().tolist(), pscore[0, :T].cpu().tolist()) if int(ok[0]) else (T, None, None))
            return out

        audio = np.load(os.path.join(GOLDEN, "logmel.npz"))["audio_sample_i16"].astype(np.float32) / 32768.0
        meta = {"language": "en", "dictionary": docs["dictionary"], "type": "hip"}
        spans = doc["sentence_spans"]
        res = AL.align([dict(s) for s in doc["segments_in"]], None, meta, audio, "cuda", _aligner=gpu_aligner,
                       _sentence_spans=lambda sdx, text: [tuple(x) for x in spans[sdx]])
        from tests.test_align_host import _norm
        assert _norm(res) == _norm(doc["result"]), name


def test_w2v_layernorm_stable_variant_and_big_vocab():
    """feat_extract_norm="layer" + do_stable_layer_norm (the XLSR family of alignment.py:39-74) and a
    vocabulary larger than one wave (character sets of the ja / zh models)."""
    from whisperx_mlx_amd.w2v import W2VConfig as C2
    cfg = C2(conv_dim=64, hidden=128, heads=2, layers=2, ffn=256, vocab=200, norm_mode=1, stable_ln=1)
    od = OWV.W2VDims(conv_dim=64, hidden=128, heads=2, layers=2, ffn=256, vocab=200, norm_mode=1, stable_ln=1)
    w = OWV.random_weights(od, seed=5)
    m = W2VHipModel.from_state_dict(w, cfg)
    waves = [synth_audio(11, 20000), synth_audio(12, 7777)]
    logp, T = m.emissions(waves)
    torch.cuda.synchronize()
    for i, wv in enumerate(waves):
        ref = OWV.emissions(w, od, torch.from_numpy(wv))
        assert T[i] == ref.shape[0]
        got = logp[i, : T[i]].cpu()
        assert torch.isfinite(got).all()
        assert (got - ref).abs().max().item() < EMIS_TOL, i


def test_w2v_from_transformers_checkpoint_dir(tmp_path):
    """`load_align_model`'s path: a transformers Wav2Vec2ForCTC directory (config.json, model.safetensors with the
    weight-norm parametrisation of the positional conv, vocab.json) loaded by W2VHipModel.from_hf_dir, against the
    transformers model's own log-softmaxed logits."""
    import json
    import warnings
    from transformers import Wav2Vec2Config, Wav2Vec2ForCTC
    from whisperx_mlx_amd.w2v import W2VHipModel
    warnings.filterwarnings("ignore")
    torch.manual_seed(2)
    cfg = Wav2Vec2Config(vocab_size=32, hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=3072,
                         conv_dim=(512,) * 7, conv_stride=(5, 2, 2, 2, 2, 2, 2), conv_kernel=(10, 3, 3, 3, 3, 2, 2),
                         feat_extract_norm="group", do_stable_layer_norm=False, num_conv_pos_embeddings=128,
                         num_conv_pos_embedding_groups=16, hidden_dropout=0.0, attention_dropout=0.0, feat_proj_dropout=0.0,
                         final_dropout=0.0, layerdrop=0.0, mask_time_prob=0.0)
    m = Wav2Vec2ForCTC(cfg).eval().half().float()
    m.save_pretrained(tmp_path, safe_serialization=True)
    vocab = {"<pad>": 0, "<s>": 1, "</s>": 2, "<unk>": 3, "|": 4, **{chr(ord("A") + i): 5 + i for i in range(26)}, "'": 31}
    json.dump(vocab, open(tmp_path / "vocab.json", "w"))
    model, v2 = W2VHipModel.from_hf_dir(str(tmp_path))
    assert v2 == vocab
    wave = [speechlike_audio(1.7, seed=8).astype(np.float32), speechlike_audio(0.9, seed=9).astype(np.float32)]
    logp, T = model.emissions(wave)
    torch.cuda.synchronize()
    for i, w in enumerate(wave):
        with torch.no_grad():
            ref = torch.log_softmax(m(torch.from_numpy(w)[None]).logits, -1)[0]
        assert T[i] == ref.shape[0]
        got = logp[i, : T[i]].cpu()
        assert float((got - ref).abs().max()) < 5e-2
        assert (got.argmax(-1) == ref.argmax(-1)).float().mean() > 0.9


def test_w2v_long_segments_take_the_256_tile_gemm():
    """>= 512 frames per segment: the conv stack, projections and FFNs run on the 256 x 256 GEMM kernel (batched over
    segments, strided conv rows, transposed V^T output with a ragged last column group) -- same oracle, same tolerance."""
    m, w = _model()
    waves = [synth_audio(21, 16000 * 12), synth_audio(22, 16000 * 11 + 137)]
    logp, T = m.emissions(waves)
    torch.cuda.synchronize()
    assert T[0] >= 512 and T[1] >= 512
    logp = logp.cpu()
    for i, wv in enumerate(waves):
        ref = OWV.emissions(w, ODIMS, torch.from_numpy(wv))
        assert T[i] == ref.shape[0]
        got = logp[i, : T[i]]
        assert torch.isfinite(got).all()
        assert (got - ref).abs().max().item() < EMIS_TOL, i


def test_w2v_base_full_depth_vs_oracle():
    """config 4's align model at its real size: wav2vec2-base (512-channel feature encoder, 768 wide, 12 heads, 12
    post-LN layers, FFN 3072, 32 labels) with seeded random weights, a ragged batch of four, against the fp32 oracle."""
    dims = OWV.W2VDims()                      # the base architecture
    cfg = W2VConfig()
    assert (cfg.hidden, cfg.layers, cfg.heads, cfg.ffn, cfg.conv_dim) == (768, 12, 12, 3072, 512)
    w = OWV.random_weights(dims, seed=11)
    m = W2VHipModel.from_state_dict(w, cfg)
    waves = [speechlike_audio(4.0, seed=1), speechlike_audio(1.3, seed=2), synth_audio(3, 40000), speechlike_audio(6.5, seed=4)]
    logp, T = m.emissions(waves)
    torch.cuda.synchronize()
    logp = logp.cpu()
    for i, wv in enumerate(waves):
        ref = OWV.emissions(w, dims, torch.from_numpy(wv))
        assert T[i] == ref.shape[0]
        got = logp[i, : T[i]]
        assert torch.isfinite(got).all()
        # 12 layers of fp16 activations against fp32: looser than the 2-layer bound, still far below a label's margin
        assert (got - ref).abs().max().item() < 6e-2, (i, (got - ref).abs().max().item())
        assert (got.argmax(-1) == ref.argmax(-1)).float().mean().item() > 0.9
