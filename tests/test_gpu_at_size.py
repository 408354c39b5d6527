"""-m gpu: BASELINE.json's configs 4 and 5 at their single-GPU size, through the public API.

config 5  distil-large-v3 (2 decoder layers) with int8 decoder weights, one hour AND the config's full ten hours of synthetic long-form audio through
          batch_processor.batch_transcribe (reference whisperx/batch_processor.py:279-338): 123 chunks of 30 s with 0.5 s
          overlap, cut by the default scheduler into wide passes.  Every chunk's tokens must equal a direct decode of the
          same chunk under a different schedule, and sampled chunks are held against the oracle step by step on the
          quantise -> dequantise checkpoint.
config 4  whisper-large-v3 (32 + 32 layers, bench weights) + wav2vec2-base + VAD-shaped chunks, transcribe ->
          align_words=True end to end (reference whisperx/asr.py:50-87, backends/mlx_lightning.py:290-369,
          alignment.py:113-380): the word dicts the pipeline returns against the same host assembly driven by the ORACLE's
          wav2vec2 emissions and DP on the same weights."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ctc as OC                   # noqa: E402
from oracle import decoding as OD              # noqa: E402
from oracle import quant as OQ                 # noqa: E402
from oracle import wav2vec2_ref as OWV         # noqa: E402
from tests import parity as PAR                # noqa: E402
from whisperx_mlx_amd import alignment as AL   # noqa: E402
from whisperx_mlx_amd import backend as BK     # noqa: E402
from whisperx_mlx_amd import weights as WT     # noqa: E402
from whisperx_mlx_amd.batch_processor import BatchProcessor, batch_transcribe   # noqa: E402
from whisperx_mlx_amd.synth import speechlike_audio                             # noqa: E402
from whisperx_mlx_amd.vad import SileroVad, merge_chunks                        # noqa: E402
from whisperx_mlx_amd.w2v import W2VConfig, W2VHipModel                         # noqa: E402


@pytest.mark.parametrize("hours", [1, 10])
def test_config5_longform_distil_int8(hours):
    """hours = 10: BASELINE.json config 5 at its full single-GPU size (1 221 chunks through batch_processor, the default
    scheduler's twelve wide passes) -- every chunk's tokens and log-probability against a direct decode under another
    schedule; hours = 1 additionally holds sampled chunks against the oracle"""
    be = BK.WhisperHipBackend("distil-large-v3", random_init=True, seed=5, compute_type="int8", max_batch=16,
                              init_std=0.03, init_emb_std=0.03)
    assert be.dims.n_text_layer == 2 and be.dims.n_audio_layer == 32 and any(k.endswith(".wq") for k in be.engine.packed)
    ten_min = speechlike_audio(600.0, seed=77)
    audio = np.tile(ten_min, 6 * hours)
    segs = [{"start": 0.0, "end": 3600.0 * hours}]
    out = batch_transcribe(audio, segs, be, batch_size=16, decode_options={"language": "en"})
    plan = dict(be.last_plan)
    chunks = BatchProcessor(batch_size=16).create_chunks(audio, segs)
    if hours == 1:
        assert len(chunks) == 123 and plan["rows"] == [43, 48, 32] and plan["passes_in_flight"] == 3      # 8 row groups: 3 + 3 + 2, the ragged one first
    else:
        assert len(chunks) == 1221 and plan["rows"] == [101, 112, 112, 112, 112] + [96] * 7 and plan["passes_in_flight"] == 3
    assert len(out) == 1 and out[0]["start"] == 0.0 and out[0]["end"] == 3600.0 * hours and out[0]["text"]
    # the scheduler's result for every chunk == a direct decode of the chunk list, one 16-row pass after the other
    sched = be._decode_chunks([c.audio for c in chunks], "en", "transcribe", False)
    direct = be._decode_chunks([c.audio for c in chunks], "en", "transcribe", False, rows_per_pass=16, passes_in_flight=1)
    assert [r["tokens"] for r in sched] == [r["tokens"] for r in direct]
    assert [r["sum_logprob"] for r in sched] == [r["sum_logprob"] for r in direct]
    merged = BatchProcessor()._merge_overlapping_text(list(zip(chunks, [{"text": r["text"]} for r in direct])))
    assert out[0]["text"] == merged
    be.engine.check_status()
    if hours != 1:
        return
    # sampled chunks against the oracle (fp32 decoder on the dequantised weights), every step
    dims, tok = be.dims, be.tokenizer
    ck = WT.random_checkpoint(dims, seed=5, std=0.03, emb_std=0.03, device="cuda")
    ckq = OQ.dequantized_checkpoint({k: v.float().cpu() for k, v in ck.items()}, dims.n_text_layer)
    sp = OD.Specials.for_vocab(dims.n_vocab)
    pick = [0, 61, 122]                                            # first, middle, last (the ragged 11.5 s tail)
    eng = be.engine
    pcm = torch.zeros(len(pick), 480000)
    lens = []
    for i, j in enumerate(pick):
        a = chunks[j].audio
        pcm[i, : len(a)] = torch.from_numpy(np.asarray(a, dtype=np.float32))
        lens.append(len(a))
    enc = eng.encode(eng.logmel(pcm.cuda(), torch.tensor(lens, dtype=torch.int32).cuda()))
    prompt = tok.sot_sequence("en", "transcribe")
    n_s = min(max(len(direct[j]["tokens"]) for j in pick) + 1, dims.n_text_ctx // 2)
    gpu = np.full((len(pick), len(prompt) + n_s), tok.eot, dtype=np.int64)
    for i, j in enumerate(pick):
        gpu[i, : len(prompt)] = prompt
        gpu[i, len(prompt): len(prompt) + len(direct[j]["tokens"])] = direct[j]["tokens"]
    rep = PAR.check_tokens_strict(ckq, dims, enc, gpu, len(prompt), n_s, sp, be.rules, be.suppress, tol=1e-2,
                                  gpu_sum_logprob=[direct[j]["sum_logprob"] for j in pick], lp_tol=0.01)
    PAR.assert_strict(rep)
    assert rep.steps_checked >= 3 * 8


TURNS = [(0.6, 7.4), (8.1, 19.0), (19.6, 28.2), (30.0, 44.5), (45.0, 52.3), (55.0, 84.9), (86.0, 101.0), (102.2, 119.5),
         (121.0, 149.0), (150.5, 176.0), (178.0, 180.9), (183.0, 212.0), (214.0, 239.5)]


def _fake_silero(wav, model=None, sampling_rate=16000, max_speech_duration_s=30, threshold=0.5):
    return [{"start": int(s * 16000), "end": int(e * 16000)} for s, e in TURNS if e * 16000 <= wav.shape[0]]


def test_config4_large_v3_vad_chunks_transcribe_then_align_end_to_end():
    be = BK.WhisperHipBackend("large-v3", random_init=True, seed=0, max_batch=16)           # bench.py's weights
    assert be.dims.n_text_layer == 32 and be.dims.n_audio_layer == 32
    # wav2vec2-base (12 layers, 768 wide), seeded weights with a sharpened CTC head: random weights give flat emissions
    # whose DP decisions are near-ties; the sharpened head makes the comparison below decisive
    odims = OWV.W2VDims()
    w = OWV.random_weights(odims, seed=11)
    w["lm_head.weight"] = (w["lm_head.weight"] * 6.0).half().float()
    cfg = W2VConfig()
    m = W2VHipModel.from_state_dict(w, cfg)
    labels = ["<pad>", "<s>", "</s>", "<unk>", "|"] + list("etaonihsrdlumwcfgypbvk'xjqz")
    meta = {"language": "en", "dictionary": {c.lower(): i for i, c in enumerate(labels)}, "type": "hip"}
    be.align_model_cache["align_en"] = (m, meta)
    pipe = BK.HipWhisperPipeline(be, SileroVad(_fake_silero, vad_onset=0.5))
    audio = speechlike_audio(240.0, seed=31)
    vad = merge_chunks(TURNS, 30)
    assert len(vad) >= 8 and all(c["end"] - c["start"] <= 30.0 for c in vad)
    # (1) the pipeline end to end on the GPU: VAD chunks -> large-v3 -> wav2vec2 forward + CTC alignment -> word dicts
    got = pipe.transcribe(audio, batch_size=16, language="en", word_timestamps=True, forced_len=40)
    be.engine.check_status()
    assert got["segments"] and all("words" in s for s in got["segments"])
    # (2) the same transcript through the same host assembly, emissions and DP from the ORACLE (fp32 CPU wav2vec2)
    segs = [dict(c, audio=audio[int(c["start"] * 16000): int(c["end"] * 16000)]) for c in vad]
    plain = be.transcribe_batch(segs, batch_size=16, language="en", forced_len=40)
    assert [s["text"] for s in plain["segments"]] == [s["text"] for s in got["segments"]]

    def oracle_aligner(waveforms, token_lists, blank_id, beam):
        out = []
        for wav, toks in zip(waveforms, token_lists):
            em = OWV.emissions(w, odims, torch.from_numpy(np.asarray(wav, dtype=np.float32))).numpy()
            tr = OC.get_trellis(em, toks, blank_id)
            path = OC.backtrack_beam(tr, em, toks, blank_id, beam)
            out.append((em.shape[0], None, None) if path is None else (em.shape[0], [p[0] for p in path], [p[2] for p in path]))
        return out

    real_align = AL.align_batch            # (_align_batch_words aligns every VAD segment's transcript in one align_batch call)
    calls = []
    try:
        def with_oracle(items, model, md, dev, **kw):
            calls.append(len(items))
            return real_align(items, None, md, "cpu", _aligner=oracle_aligner, **kw)
        AL.align_batch = with_oracle
        ref = be._align_batch_words(copy.deepcopy(plain), segs)
    finally:
        AL.align_batch = real_align
    assert calls == [len(plain["segments"])]          # the oracle really drove the comparison run
    assert len(ref["segments"]) == len(got["segments"])
    n_words = n_timed = n_close = 0
    for a, b in zip(got["segments"], ref["segments"]):
        assert a["text"] == b["text"] and abs(a["start"] - b["start"]) < 1e-6 and abs(a["end"] - b["end"]) < 1e-6
        assert [x["word"] for x in a["words"]] == [x["word"] for x in b["words"]]
        for x, y in zip(a["words"], b["words"]):
            n_words += 1
            assert ("start" in x) == ("start" in y)
            if "start" in y:
                n_timed += 1
                n_close += abs(x["start"] - y["start"]) <= 0.0201 and abs(x["end"] - y["end"]) <= 0.0201
    assert n_words >= 100 and n_timed >= 50
    # fp16 activations through 12 layers against the fp32 oracle move a handful of near-tied DP decisions even with the
    # sharpened head; every other word must sit within +-20 ms at both ends
    assert n_close >= 0.97 * n_timed, (n_close, n_timed, n_words)
    # (3) config 4's multi-GPU entry point at world size 1 (no process group): transcribe + align of the rank's share, the
    # aligned words through the fixed-width record (pack -> unpack -> assemble), must give the pipeline's dict exactly
    from whisperx_mlx_amd import parallel as PL
    sharded = PL.transcribe_batch_sharded(be, segs, batch_size=16, align_words=True, language="en", forced_len=40)
    assert sharded["language"] == got["language"] and len(sharded["segments"]) == len(got["segments"])
    for a, b in zip(sharded["segments"], got["segments"]):
        assert a == b
