"""-m gpu: log-mel and CTC alignment kernels against the golden vectors produced by the
reference's own code (tests/golden, tools/make_golden.py) and against the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import gpu_util as G   # noqa: E402
from tests.conftest import GOLDEN  # noqa: E402
from oracle import ctc as OC       # noqa: E402
from oracle import logmel as OL    # noqa: E402
from whisperx_mlx_amd import weights  # noqa: E402
from whisperx_mlx_amd.audio import mel_filters  # noqa: E402

LOGMEL_TOL = 2e-4   # abs, on the (log10 + 4) / 4 scale; reference itself is fp32 FFT


def _engine128():
    """n_mels = 128 needs its own context (filterbank is per model)."""
    dims = weights.ModelDimensions(128, 1500, 128, 2, 1, 51866, 448, 128, 2, 1)
    if not hasattr(_engine128, "e"):
        ck = weights.random_checkpoint(dims, seed=1)
        from whisperx_mlx_amd.engine import WhisperHipEngine
        _engine128.e = WhisperHipEngine(dims, weights.pack(ck, dims, "cuda"), max_batch=4)
    return _engine128.e


def _mel(eng, chunks):
    B = len(chunks)
    n = max(len(c) for c in chunks)
    pcm = torch.zeros(B, max(n, 1), dtype=torch.float32)
    nv = torch.zeros(B, dtype=torch.int32)
    for i, c in enumerate(chunks):
        pcm[i, : len(c)] = torch.from_numpy(np.asarray(c, dtype=np.float32))
        nv[i] = len(c)
    mel16, mel32 = eng.logmel(pcm.cuda(), nv.cuda(), want_f32=True)
    torch.cuda.synchronize()
    return mel16.float().cpu().numpy(), mel32.cpu().numpy()


def test_logmel_golden_sample():
    g = np.load(os.path.join(GOLDEN, "logmel.npz"))
    a = g["audio_sample_i16"].astype(np.float32) / 32768.0
    for n_mels, eng in ((80, G.tiny_engine()[0]), (128, _engine128())):
        _, m = _mel(eng, [a])
        m = m[0].T                                    # (n_mels, 3000)
        assert np.abs(m[:, :64] - g[f"chunk_mel{n_mels}_head"]).max() < LOGMEL_TOL
        assert np.abs(m[:, 468:532] - g[f"chunk_mel{n_mels}_mid"]).max() < LOGMEL_TOL
        assert np.abs(m[:, -64:] - g[f"chunk_mel{n_mels}_tail"]).max() < LOGMEL_TOL
        st = g[f"chunk_mel{n_mels}_stats"]
        assert abs(m.mean(dtype=np.float64) - st[0]) < 1e-5 and abs(m.min() - st[1]) < LOGMEL_TOL
        assert abs(m.max() - st[2]) < LOGMEL_TOL


def test_logmel_golden_synth_ragged_batch():
    """ragged lengths in one batch: 1 s, odd length, exactly 30 s, 1000 samples"""
    g = np.load(os.path.join(GOLDEN, "logmel.npz"))
    from tests.synth import synth_audio
    chunks = [synth_audio(seed, int(g[f"synth{seed}_n"][0])) for seed in (1, 2, 3, 4)]
    m16, m32 = _mel(_engine128(), chunks)
    for i, seed in enumerate((1, 2, 3, 4)):
        m = m32[i].T
        lo = int(g[f"synth{seed}_edge_lo"][0])
        assert np.abs(m[:, :48] - g[f"synth{seed}_head"]).max() < LOGMEL_TOL
        assert np.abs(m[:, lo:lo + 48] - g[f"synth{seed}_edge"]).max() < LOGMEL_TOL
        assert np.abs(m[:, -48:] - g[f"synth{seed}_tail"]).max() < LOGMEL_TOL
        assert abs(m.mean(dtype=np.float64) - g[f"synth{seed}_stats"][0]) < 1e-5
        assert np.abs(m16[i].T - m).max() < 2e-3      # fp16 copy of the same values


def test_logmel_vs_oracle_random():
    rng = np.random.default_rng(7)
    chunks = [rng.normal(0, 0.1, n).astype(np.float32) for n in (480000, 123457, 3)]
    chunks.append(np.zeros(5000, dtype=np.float32))     # silence: everything clamps
    eng = G.tiny_engine()[0]
    _, m32 = _mel(eng, chunks)
    ref = OL.log_mel_chunks(chunks, [len(c) for c in chunks], mel_filters(80))
    assert np.abs(m32 - ref).max() < LOGMEL_TOL


CASES = ["wild", "single", "two", "tight", "toolong", "long", "blank5"]


def _run_ctc(eng, names, c, beam=2):
    S = len(names)
    Tmax = max(c[n + "_emission"].shape[0] for n in names)
    V = max(c[n + "_emission"].shape[1] for n in names)
    Nmax = max(len(c[n + "_tokens"]) for n in names)
    logp = torch.full((S, Tmax, V), -1e30, dtype=torch.float32)
    tok = torch.zeros(S, Nmax, dtype=torch.int32)
    T = torch.zeros(S, dtype=torch.int32)
    N = torch.zeros(S, dtype=torch.int32)
    for i, n in enumerate(names):
        e = c[n + "_emission"]
        logp[i, : e.shape[0], : e.shape[1]] = torch.from_numpy(e)
        tok[i, : len(c[n + "_tokens"])] = torch.from_numpy(c[n + "_tokens"])
        T[i], N[i] = e.shape[0], len(c[n + "_tokens"])
    return eng.ctc_align(logp, T, tok, N, blank_id=int(c[names[0] + "_blank"][0]), beam=beam, want_trellis=True)


def test_ctc_golden_bit_exact():
    """trellis bit-exact, beam path token/time indices identical, scores to 1 ulp of exp."""
    c = np.load(os.path.join(GOLDEN, "ctc.npz"))
    eng = G.tiny_engine()[0]
    for names in (["wild", "two", "long"], ["single", "tight", "toolong"], ["blank5"]):
        # a batch shares V and blank id
        ptok, pscore, ok, tr = _run_ctc(eng, names, c)
        torch.cuda.synchronize()
        ptok, pscore, ok, tr = ptok.cpu().numpy(), pscore.cpu().numpy(), ok.cpu().numpy(), tr.cpu().numpy()
        for i, n in enumerate(names):
            Tn, Nn = c[n + "_emission"].shape[0], len(c[n + "_tokens"])
            assert np.array_equal(tr[i, :Tn, :Nn], c[n + "_trellis"]), n
            assert int(ok[i]) == int(c[n + "_ok"][0]), n
            if ok[i]:
                assert np.array_equal(ptok[i, :Tn], c[n + "_path_tok"]), n
                assert np.array_equal(np.arange(Tn), c[n + "_path_time"]), n
                assert np.abs(pscore[i, :Tn] - c[n + "_path_score"]).max() < 1e-6, n


def test_ctc_beam5_and_oracle_random():
    c = np.load(os.path.join(GOLDEN, "ctc.npz"))
    eng = G.tiny_engine()[0]
    ptok, _, ok, _ = _run_ctc(eng, ["wild", "two", "long"], c, beam=5)
    for i, n in enumerate(["wild", "two", "long"]):
        Tn = c[n + "_emission"].shape[0]
        assert np.array_equal(ptok[i, :Tn].cpu().numpy(), c[n + "_path5_tok"]), n
    # seeded random batch against the oracle, ragged T and N
    rng = np.random.default_rng(3)
    S, V = 6, 29
    Ts = [50, 300, 1499, 7, 120, 64]
    Ns = [10, 80, 400, 7, 1, 30]
    logp = torch.full((S, max(Ts), V), 0.0)
    tok = torch.zeros(S, max(Ns), dtype=torch.int32)
    refs = []
    for i in range(S):
        e = torch.log_softmax(torch.from_numpy(rng.normal(0, 2, (Ts[i], V)).astype(np.float32)), -1).numpy()
        t = rng.integers(1, V, Ns[i]).tolist()
        if Ns[i] > 3:
            t[2] = -1
        logp[i, : Ts[i]] = torch.from_numpy(e)
        tok[i, : Ns[i]] = torch.tensor(t, dtype=torch.int32)
        tr = OC.get_trellis(e, t, 0)
        refs.append((tr, OC.backtrack_beam(tr, e, t, 0, 2)))
    ptok, pscore, ok, tr = eng.ctc_align(logp, torch.tensor(Ts), tok, torch.tensor(Ns), 0, 2, want_trellis=True)
    torch.cuda.synchronize()
    for i in range(S):
        assert np.array_equal(tr[i, : Ts[i], : Ns[i]].cpu().numpy(), refs[i][0]), i
        if refs[i][1] is None:
            assert int(ok[i]) == 0
        else:
            assert int(ok[i]) == 1
            assert ptok[i, : Ts[i]].cpu().tolist() == [q[0] for q in refs[i][1]], i


@pytest.mark.parametrize("V", [200, 2500])
def test_ctc_large_vocab_staging_variants(V):
    """emission chunks of fewer than 32 frames (V = 200) and no LDS staging at all (V = 2500): same trellis and
    path as the oracle, bit for bit"""
    eng = G.tiny_engine()[0]
    rng = np.random.default_rng(V)
    Ts, Ns = [97, 33, 260], [20, 33, 75]
    S = len(Ts)
    logp = torch.zeros(S, max(Ts), V)
    tok = torch.zeros(S, max(Ns), dtype=torch.int32)
    refs = []
    for i in range(S):
        e = torch.log_softmax(torch.from_numpy(rng.normal(0, 2, (Ts[i], V)).astype(np.float32)), -1).numpy()
        t = rng.integers(1, V, Ns[i]).tolist()
        t[1] = -1
        logp[i, : Ts[i]] = torch.from_numpy(e)
        tok[i, : Ns[i]] = torch.tensor(t, dtype=torch.int32)
        tr = OC.get_trellis(e, t, 0)
        refs.append((tr, OC.backtrack_beam(tr, e, t, 0, 2)))
    ptok, pscore, ok, tr = eng.ctc_align(logp, torch.tensor(Ts), tok, torch.tensor(Ns), 0, 2, want_trellis=True)
    torch.cuda.synchronize()
    for i in range(S):
        assert np.array_equal(tr[i, : Ts[i], : Ns[i]].cpu().numpy(), refs[i][0]), i
        if refs[i][1] is None:
            assert int(ok[i]) == 0
        else:
            assert int(ok[i]) == 1
            assert ptok[i, : Ts[i]].cpu().tolist() == [q[0] for q in refs[i][1]], i


def test_ctc_wide_beam_long_segment_records_in_hbm():
    """beam 8 over 2200 frames: the per-step beam records no longer fit in LDS and go through the HBM scratch;
    the windowed backtrack must still follow the oracle exactly (also with beams spread over many columns)"""
    eng = G.tiny_engine()[0]
    rng = np.random.default_rng(77)
    T, N, V = 2200, 300, 32
    e = torch.log_softmax(torch.from_numpy(rng.normal(0, 2, (T, V)).astype(np.float32)), -1).numpy()
    t = rng.integers(1, V, N).tolist()
    t[5] = -1
    tr = OC.get_trellis(e, t, 0)
    ref = OC.backtrack_beam(tr, e, t, 0, 8)
    ptok, pscore, ok, trg = eng.ctc_align(torch.from_numpy(e)[None], torch.tensor([T]), torch.tensor([t], dtype=torch.int32),
                                          torch.tensor([N]), 0, 8, want_trellis=True)
    torch.cuda.synchronize()
    assert np.array_equal(trg[0, :T, :N].cpu().numpy(), tr)
    assert ref is not None and int(ok[0]) == 1
    assert ptok[0, :T].cpu().tolist() == [q[0] for q in ref]
    assert np.abs(pscore[0, :T].cpu().numpy() - np.array([q[2] for q in ref], np.float32)).max() < 1e-6


def test_ctc_very_long_segment_takes_the_plain_kernel():
    """13000 frames (4.3 min in one segment): the per-frame arrays exceed LDS, the plain kernel must give the
    oracle's trellis and the greedy-equivalent beam-1 path"""
    eng = G.tiny_engine()[0]
    rng = np.random.default_rng(5)
    T, N, V = 13000, 120, 29
    e = torch.log_softmax(torch.from_numpy(rng.normal(0, 2, (T, V)).astype(np.float32)), -1).numpy()
    t = rng.integers(1, V, N).tolist()
    tr = OC.get_trellis(e, t, 0)
    ptok, pscore, ok, trg = eng.ctc_align(torch.from_numpy(e)[None], torch.tensor([T]), torch.tensor([t], dtype=torch.int32),
                                          torch.tensor([N]), 0, 1, want_trellis=True)
    torch.cuda.synchronize()
    assert np.array_equal(trg[0, :T, :N].cpu().numpy(), tr)
    assert int(ok[0]) == 1
    path = ptok[0, :T].cpu().numpy()
    assert path[0] == 0 and path[-1] == N - 1 and (np.diff(path) >= 0).all() and (np.diff(path) <= 1).all()
    # beam 1 = follow the better of stay / change at every frame (alignment.py:531-565 with one survivor)
    j = N - 1
    for tt in range(T - 1, 0, -1):
        assert path[tt] == j
        if j == 0:
            break
        stay, change = tr[tt - 1, j], tr[tt - 1, j - 1]
        cands = [(s, c) for s, c in ((stay, j), (change, j - 1)) if np.isfinite(s)]
        j = max(cands, key=lambda x: x[0])[1] if cands[0][0] != cands[-1][0] else cands[0][1]
