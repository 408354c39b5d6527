"""-m gpu: config 4's front end -- Silero-style VAD turns -> merge_chunks -> ragged (<= 30 s) chunks -> the hot path --
with ORACLE parity on every ragged chunk (log-mel, encoder, greedy tokens step by step), not only result-shape
properties.  The Silero network is a CPU pre-step outside the hot path (reference: whisperx/vads/silero.py:30-49); a
deterministic stand-in with its `get_speech_timestamps` signature produces the turns here (no hub access on the box)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import decoding as OD      # noqa: E402
from oracle import logmel as OL        # noqa: E402
from oracle import whisper_ref as OW   # noqa: E402
from tests import gpu_util as G        # noqa: E402
from tests import parity as PAR        # noqa: E402
from tests.synth import speechlike_audio           # noqa: E402
from whisperx_mlx_amd import backend as BK         # noqa: E402
from whisperx_mlx_amd import weights as WT         # noqa: E402
from whisperx_mlx_amd.audio import mel_filters     # noqa: E402
from whisperx_mlx_amd.vad import SileroVad, merge_chunks   # noqa: E402

TURNS = [(0.8, 6.1), (6.9, 14.0), (15.2, 27.9), (29.0, 41.5), (43.0, 44.2), (50.0, 79.0), (80.5, 94.0)]


def _fake_silero(wav, model=None, sampling_rate=16000, max_speech_duration_s=30, threshold=0.5):
    assert sampling_rate == 16000 and wav.dim() == 1 and 0 < threshold < 1
    return [{"start": int(s * 16000), "end": int(e * 16000)} for s, e in TURNS if e * 16000 <= wav.shape[0]]


def test_merge_chunks_semantics():
    # vads/vad.py:20-53 on the turns above: a turn that would push the running chunk past chunk_size opens a new one
    got = [(c["start"], c["end"], len(c["segments"])) for c in merge_chunks(TURNS, 30)]
    assert got == [(0.8, 27.9, 3), (29.0, 44.2, 2), (50.0, 79.0, 1), (80.5, 94.0, 1)]
    assert merge_chunks([], 30) == []


def test_vad_chunks_through_the_hot_path_against_oracle():
    be = BK.WhisperHipBackend("tiny", random_init=True, seed=7, init_std=0.1, init_emb_std=0.06, max_batch=8)
    pipe = BK.HipWhisperPipeline(be, SileroVad(_fake_silero, vad_onset=0.5))
    audio = speechlike_audio(95.0, seed=21)
    res = pipe.transcribe(audio, batch_size=8, language="en", return_chunks=True)
    chunks = merge_chunks(TURNS, 30)
    assert len(res["chunks"]) == len(chunks) == 4
    for seg in res["segments"]:
        assert any(c["start"] - 1e-6 <= seg["start"] and seg["end"] <= c["end"] + 1e-6 for c in chunks)
    # ---- oracle on the same ragged chunks
    dims = be.dims
    ck = {k: v.float().cpu() for k, v in WT.random_checkpoint(dims, seed=7, std=0.1, emb_std=0.06, device="cuda").items()}
    pcm = [audio[int(c["start"] * 16000): int(c["end"] * 16000)] for c in chunks]
    lens = [len(p) for p in pcm]
    assert len(set(lens)) == 4 and max(lens) < 480000          # ragged, none a full window
    eng, tok = be.engine, be.tokenizer
    dev = torch.zeros(4, 480000)
    for i, p in enumerate(pcm):
        dev[i, : len(p)] = torch.from_numpy(p)
    mel16, mel32 = eng.logmel(dev.cuda(), torch.tensor(lens, dtype=torch.int32).cuda(), want_f32=True)
    ref_mel = OL.log_mel_chunks(pcm, lens, mel_filters(dims.n_mels))
    assert np.abs(mel32.cpu().numpy() - ref_mel).max() < 2e-4
    enc = eng.encode(mel16)
    ref_enc = OW.encoder_forward(ck, dims, torch.from_numpy(ref_mel))
    for i in range(4):
        assert G.rel_err(enc[i], ref_enc[i]) < 2e-2, i
    # tokens the pipeline returned for every chunk: each step against the oracle teacher-forced along them
    sp = OD.Specials.for_vocab(dims.n_vocab)
    prompt = tok.sot_sequence("en", "transcribe")
    n_s = min(max(len(c["tokens"]) for c in res["chunks"]) + 1, dims.n_text_ctx // 2)      # + the EOT step, up to sample_len
    gpu = np.full((4, len(prompt) + n_s), tok.eot, dtype=np.int64)
    for i, c in enumerate(res["chunks"]):
        gpu[i, : len(prompt)] = prompt
        gpu[i, len(prompt): len(prompt) + len(c["tokens"])] = c["tokens"]
    rep = PAR.check_tokens_strict(ck, dims, enc, gpu, len(prompt), n_s, sp, be.rules, be.suppress, tol=1e-2,
                                  gpu_sum_logprob=[c["sum_logprob"] for c in res["chunks"]], lp_tol=0.01)
    PAR.assert_strict(rep)
    assert rep.steps_checked >= 8
    eng.check_status()
