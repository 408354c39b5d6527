"""CPU: the oracle against the golden vectors produced by the reference's own code
(tools/make_golden.py) -- this is what pins the oracle (oracle/__init__.py)."""
import os

import numpy as np
import pytest

from oracle import ctc as OC
from oracle import logmel as OL
from tests.conftest import GOLDEN
from tests.synth import synth_audio
from whisperx_mlx_amd.audio import mel_filters


def test_mel_filterbank_matches_reference_asset():
    ref = np.load(os.path.join(GOLDEN, "mel_filters_ref.npz"))
    for n in (80, 128):
        ours = mel_filters(n)
        assert ours.shape == (n, 201) and ours.dtype == np.float32
        assert np.abs(ours - ref[f"mel_{n}"]).max() < 1e-8
        assert np.array_equal(ours != 0, ref[f"mel_{n}"] != 0)


def test_logmel_oracle_vs_reference_whole_clip_and_chunk():
    g = np.load(os.path.join(GOLDEN, "logmel.npz"))
    a = g["audio_sample_i16"].astype(np.float32) / 32768.0
    for n in (80, 128):
        f = mel_filters(n)
        m = OL.log_mel_spectrogram(a, f)
        assert m.shape == (n, 500)
        assert np.abs(m - g[f"sample_mel{n}"]).max() < 1e-4
        c = OL.log_mel_spectrogram(OL.pad_or_trim(a), f)
        assert np.abs(c[:, :64] - g[f"chunk_mel{n}_head"]).max() < 1e-4
        assert np.abs(c[:, 468:532] - g[f"chunk_mel{n}_mid"]).max() < 1e-4
        assert np.abs(c[:, -64:] - g[f"chunk_mel{n}_tail"]).max() < 1e-4
        st = g[f"chunk_mel{n}_stats"]
        assert abs(c.mean(dtype=np.float64) - st[0]) < 1e-6 and abs(c.max() - st[2]) < 1e-4


def test_logmel_oracle_vs_reference_synthetic_ragged():
    g = np.load(os.path.join(GOLDEN, "logmel.npz"))
    f = mel_filters(128)
    for seed in (1, 2, 4):
        n = int(g[f"synth{seed}_n"][0])
        m = OL.log_mel_chunks([synth_audio(seed, n)], [n], f)[0].T
        lo = int(g[f"synth{seed}_edge_lo"][0])
        assert np.abs(m[:, :48] - g[f"synth{seed}_head"]).max() < 1e-4
        assert np.abs(m[:, lo:lo + 48] - g[f"synth{seed}_edge"]).max() < 1e-4
        assert np.abs(m[:, -48:] - g[f"synth{seed}_tail"]).max() < 1e-4


@pytest.mark.parametrize("name", ["wild", "single", "two", "tight", "toolong", "long", "blank5"])
def test_ctc_oracle_vs_reference(name):
    c = np.load(os.path.join(GOLDEN, "ctc.npz"))
    em, tok, blank = c[name + "_emission"], c[name + "_tokens"].tolist(), int(c[name + "_blank"][0])
    tr = OC.get_trellis(em, tok, blank)
    assert np.array_equal(tr, c[name + "_trellis"])          # bit-exact float32
    path = OC.backtrack_beam(tr, em, tok, blank, 2)
    if not int(c[name + "_ok"][0]):
        assert path is None
        return
    assert [p[0] for p in path] == c[name + "_path_tok"].tolist()
    assert [p[1] for p in path] == c[name + "_path_time"].tolist()
    assert np.abs(np.array([p[2] for p in path]) - c[name + "_path_score"]).max() < 1e-6
    text = "".join(chr(97 + (k % 26)) for k in range(len(tok)))
    segs = OC.merge_repeats(path, text)
    assert [s[1] for s in segs] == c[name + "_seg_start"].tolist()
    assert [s[2] for s in segs] == c[name + "_seg_end"].tolist()
    assert np.abs(np.array([s[3] for s in segs]) - c[name + "_seg_score"]).max() < 1e-6
    p5 = OC.backtrack_beam(tr, em, tok, blank, 5)
    assert [p[0] for p in p5] == c[name + "_path5_tok"].tolist()
