"""Pins taken from reference files that import in the build container (tools/make_golden.py, round 3):

  * /root/reference/median_filter_fix.py:6-35 (as-is)            -> tests/golden/median.npz
  * /root/reference/whisperx/vads/vad.py:20-53  (pyannote stub)  -> tests/golden/vad_merge.json
  * /root/reference/whisperx/batch_processor.py:47-148,186-276 (mlx.core stub) -> tests/golden/batch_processor.json

CPU: the oracle's median filter, `vad.merge_chunks` and `BatchProcessor` against those outputs.
-m gpu: the HIP running median (dtw.hip `median7` / `reflect`, through wx_median7_rows) against the same fixture."""
import json
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN
from oracle import dtw as OD
from whisperx_mlx_amd import vad as V
from whisperx_mlx_amd.batch_processor import BatchProcessor

MEDIAN_CASES = ["soft_40x1500", "randn_7x64", "ties_5x33", "short_3x3", "min_2x4", "one_1x1500", "cube_2x3x50"]


def test_oracle_median_filter_equals_the_reference():
    """oracle.dtw.median_filter_reflect == median_filter_fixed, bit for bit (a median selects, it does not round),
    widths 7 and 3, 2-D and 3-D inputs, ties, and the row too short to filter (returned unchanged)."""
    g = np.load(os.path.join(GOLDEN, "median.npz"))
    for name in MEDIAN_CASES:
        x = g[f"{name}_in"]
        for width in (7, 3):
            want = g[f"{name}_w{width}"]
            got = OD.median_filter_reflect(x, width)
            assert got.shape == want.shape and got.dtype == np.float32
            assert np.array_equal(got, want), (name, width)


def test_merge_chunks_equals_the_reference():
    with open(os.path.join(GOLDEN, "vad_merge.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) >= 8
    for c in cases:
        turns = [tuple(t) for t in c["turns"]]
        got = V.merge_chunks(turns, c["chunk_size"], onset=0.5, offset=0.363)
        want = c["merged"]
        assert len(got) == len(want), c["name"]
        for a, b in zip(got, want):
            assert a["start"] == b["start"] and a["end"] == b["end"], c["name"]
            assert [list(x) for x in a["segments"]] == b["segments"], c["name"]
        # every turn lands in exactly one chunk, in order
        assert [tuple(x) for m in got for x in m["segments"]] == turns


def test_silero_front_end_feeds_merge_chunks_like_the_reference():
    """SileroVad.__call__ = get_speech_timestamps (samples) -> seconds -> merge_chunks (vads/silero.py:30-66): with a
    detector that returns the fixture's turns the chunks are the reference's."""
    with open(os.path.join(GOLDEN, "vad_merge.json")) as f:
        c = json.load(f)["cases"][0]
    calls = {}

    def fake_timestamps(wav, model=None, sampling_rate=16000, max_speech_duration_s=30, threshold=0.5):
        calls.update(sr=sampling_rate, max_s=max_speech_duration_s, thr=threshold, n=len(wav))
        return [{"start": int(round(s * 16000)), "end": int(round(e * 16000))} for s, e in c["turns"]]

    vad = V.SileroVad(fake_timestamps, model=object(), vad_onset=0.5)
    got = vad(np.zeros(16000, dtype=np.float32), c["chunk_size"])
    assert calls == {"sr": 16000, "max_s": c["chunk_size"], "thr": 0.5, "n": 16000}
    assert len(got) == len(c["merged"])
    for a, b in zip(got, c["merged"]):
        assert abs(a["start"] - b["start"]) < 1e-4 and abs(a["end"] - b["end"]) < 1e-4


def test_silero_from_hub_loads_a_local_repo(tmp_path):
    """SileroVad.from_hub(repo_dir=...) goes through torch.hub's local-source path (vads/silero.py:23-28 offline): a
    directory with a hubconf.py exposing `silero_vad` -> (model, utils) is loaded and its first util is the detector."""
    (tmp_path / "hubconf.py").write_text(
        "dependencies = []\n"
        "def _ts(wav, model=None, sampling_rate=16000, max_speech_duration_s=30, threshold=0.5):\n"
        "    return [{'start': 1600, 'end': 48000}, {'start': 64000, 'end': 80000}]\n"
        "def silero_vad(onnx=False, **kw):\n"
        "    return 'MODEL', (_ts, None, None, None, None)\n")
    vad = V.SileroVad.from_hub(repo_dir=str(tmp_path), vad_onset=0.4)
    assert vad.model == "MODEL" and vad.vad_onset == 0.4
    chunks = vad(np.zeros(90000, dtype=np.float32), 30)
    assert chunks == [{"start": 0.1, "end": 5.0, "segments": [(0.1, 3.0), (4.0, 5.0)]}]
    with pytest.raises(RuntimeError, match="not available offline"):
        V.SileroVad.from_hub(repo_dir=str(tmp_path / "missing"))
    with pytest.raises(ValueError):
        V.SileroVad(lambda *a, **k: [], vad_onset=1.5)


def test_batch_processor_equals_the_reference():
    with open(os.path.join(GOLDEN, "batch_processor.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) >= 6
    for c in cases:
        p = BatchProcessor(batch_size=c["batch_size"], chunk_duration=c["chunk_duration"], overlap=c["overlap"])
        audio = np.arange(int(c["total_s"] * 16000), dtype=np.float32)
        segs = [{"start": a, "end": b} for a, b in c["segments"]]
        chunks = p.create_chunks(audio, segs)
        assert len(chunks) == len(c["chunks"]), c["name"]
        for ch, w in zip(chunks, c["chunks"]):
            assert (ch.start_time, ch.end_time, ch.segment_idx, len(ch.audio)) == (w["start"], w["end"], w["segment_idx"], w["n"]), c["name"]
            if w["n"]:
                assert float(ch.audio[0]) == w["first"] and float(ch.audio[-1]) == w["last"]
        batches = p.create_batches(chunks)
        assert [len(b) for b in batches] == c["batch_lens"]
        for b, w in zip(batches, c["pads"]):
            if w is None:
                continue
            padded, lens = p.pad_batch(b)
            assert list(padded.shape) == w["shape"] and [int(x) for x in lens] == w["lengths"]
            assert [float(r.astype(np.float64).sum()) for r in padded] == w["row_sums"]
        order = c["order"]
        merged = p.merge_results([chunks[i] for i in order], [c["results"][i] for i in order], segs)
        assert merged == c["merged"], c["name"]


@pytest.mark.gpu
def test_hip_median7_equals_the_reference():
    """the device functions the DTW pre-processing kernels call (dtw.hip), on the reference's own outputs: bit-exact"""
    import torch
    from tests import gpu_util as G
    from whisperx_mlx_amd import _lib
    eng, _ = G.tiny_engine()
    g = np.load(os.path.join(GOLDEN, "median.npz"))
    L = _lib.lib()
    for name in ("soft_40x1500", "randn_7x64", "ties_5x33", "min_2x4", "one_1x1500"):
        x = torch.from_numpy(g[f"{name}_in"]).cuda().contiguous()
        y = torch.zeros_like(x)
        rc = L.wx_median7_rows(eng.ctx, _lib.ptr(x), x.stride(0), x.shape[0], x.shape[1], _lib.ptr(y), y.stride(0), None)
        _lib.check(eng.ctx, rc, "wx_median7_rows")
        torch.cuda.synchronize()
        assert np.array_equal(y.cpu().numpy(), g[f"{name}_w7"]), name
    x = torch.zeros(2, 3, device="cuda")
    assert L.wx_median7_rows(eng.ctx, _lib.ptr(x), 3, 2, 3, _lib.ptr(x), 3, None) != 0       # a row the reflect pad does not fit
