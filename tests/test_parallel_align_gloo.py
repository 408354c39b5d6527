"""CPU: config 4's N > 1 path -- VAD segments sharded over ranks, each rank transcribes AND force-aligns its share, the
wav2vec2-aligned words travel in the same single gather -- with world_size 2 on gloo.

The real WhisperHipBackend host code runs (transcribe_batch, _group_by_vad, _align_batch_words, _offset_aligned); only
the two GPU halves are stood in for: `_decode_chunks` (tokens derived from the chunk's content, decoded through a small
vocabulary so that texts have sentences, abbreviations and out-of-dictionary characters) and the numeric aligner
(deterministic emissions from the waveform + the oracle's CTC DP, oracle/ctc.py).  Checked: the dict every rank ends with
equals the single-process `transcribe_batch(..., align_words=True)` dict -- segment texts, times, every aligned word with its
start / end / score, the failure branch of align() -- through exactly ONE collective.
Reference semantics: /root/reference/whisperx/alignment.py:206-373, /root/reference/whisperx/backends/mlx_lightning.py:290-369."""
import hashlib
import json
import os

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ctc as OC
from tests.conftest import GOLDEN
from whisperx_mlx_amd import alignment as AL
from whisperx_mlx_amd import backend as BK
from whisperx_mlx_amd import parallel as P
from whisperx_mlx_amd.tokenizer import Tokenizer

WORDS = [" Mr.", " Smith", " went", " home.", " He", " came", " back?", " Yes.", " It", " cost", " 3.5", " dollars.",
         " that's", " what", " they", " said", " naïve", " café", ",", " and", " so", " on...", " later.", " Dr.", " Who?"]
BASE = 1000


class _Tok(Tokenizer):
    """the product tokenizer with a small word vocabulary (no tokenizer.json ships with the repo)"""

    def decode(self, ids):
        return "".join(WORDS[(t - BASE) % len(WORDS)] for t in ids if t < self.eot)


def _dictionary():
    with open(os.path.join(GOLDEN, "align.json")) as f:
        return json.load(f)["dictionary"]


def _cpu_aligner(waveforms, token_lists, blank_id, beam):
    """stands in for _HipAligner: log-softmax of seeded noise with the wav2vec2 frame count of the waveform (every rank
    gets the same emissions for the same audio), then the oracle's trellis + beam backtrack"""
    out = []
    for wav, toks in zip(waveforms, token_lists):
        T = max(0, (len(wav) - 400) // 320 + 1)
        seed = int.from_bytes(hashlib.sha256(np.ascontiguousarray(wav[:4096]).tobytes()).digest()[:4], "little")
        g = np.random.default_rng(seed)
        em = g.standard_normal((max(T, 1), 32)).astype(np.float32)
        em = em - np.log(np.exp(em).sum(axis=1, keepdims=True))
        if T < 2:
            out.append((T, None, None))
            continue
        tr = OC.get_trellis(em, toks, blank_id)
        path = OC.backtrack_beam(tr, em, toks, blank_id, beam)
        out.append((T, None, None) if path is None else (T, [p[0] for p in path], [p[2] for p in path]))
    return out


class _HostBackend(BK.WhisperHipBackend):
    def __init__(self):       # no GPU: only the attributes the host code reads
        self.model_name, self.device_index = "large-v3", 0
        self.tokenizer = _Tok(n_vocab=51866)
        self.auto_rows, self.max_batch, self.coalesce = True, 16, 1
        self.decode_calls = 0

    def detect_language(self, audio):
        return "en"

    def _decode_chunks(self, chunks, language, task, word_timestamps, **kw):
        self.decode_calls += 1
        out = []
        for c in chunks:
            c = np.asarray(c)
            n = 0 if len(c) < 4000 else 2 + int(abs(float(c[0])) * 1000) % 17         # a very short chunk: empty text, no segment
            first = int(abs(float(c[1])) * 1000) % len(WORDS)
            toks = [50365] + [BASE + (first + k) % len(WORDS) for k in range(n)] + [50365 + len(c) // 320]
            text = self.tokenizer.decode([t for t in toks if t < self.tokenizer.eot]).strip()
            out.append({"tokens": toks, "text": text, "avg_logprob": -0.5, "sum_logprob": -0.5 * (len(toks) + 1),
                        "no_speech_prob": 0.25, "language": language, "compression_ratio": 1.0})
        return out

    def align_groups(self, groups, segments, language, _trace=None):
        meta = {"language": language, "dictionary": _dictionary(), "type": "hip"}
        return AL.align_batch([(rel, segments[vi]["audio"]) for vi, rel in groups], None, meta, "cpu", _aligner=_cpu_aligner,
                              _trace=_trace)


def _segments():
    rng = np.random.default_rng(11)
    segs, t = [], 0.0
    lens = [480000, 9000, 3000, 250000, 31000, 480000, 120000, 64000, 200000, 16000, 333333]     # 3000: empty text; 9000: T < tokens -> backtrack fails
    for i, n in enumerate(lens):
        a = (rng.standard_normal(n) * 0.1).astype(np.float32)
        a[0], a[1] = 0.001 * (3 + 5 * i), 0.001 * (7 * i + 1)
        if n == 9000:
            a[0] = 0.016            # 2 + 16 = 18 words on 28 frames of audio
        segs.append({"start": t, "end": t + n / 16000.0, "audio": a})
        t += n / 16000.0 + 0.37
    return segs


def _count_collectives():
    calls = {"n": 0}
    for name in ("all_gather_into_tensor", "all_gather", "all_reduce", "broadcast", "gather", "all_to_all"):
        fn = getattr(dist, name)

        def wrapped(*a, _fn=fn, **k):
            calls["n"] += 1
            return _fn(*a, **k)
        setattr(dist, name, wrapped)
    return calls


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = _count_collectives()
        be = _HostBackend()
        got = P.transcribe_batch_sharded(be, _segments(), batch_size=16, align_words=True, language=None, reuse_own=(rank == 0))
        q.put((rank, got, calls["n"], be.decode_calls))
    finally:
        dist.destroy_process_group()


def _single():
    be = _HostBackend()
    return be.transcribe_batch(_segments(), batch_size=16, align_words=True, language="en")


def test_single_process_reference_has_what_the_test_needs():
    res = _single()
    segs = res["segments"]
    words = [w for s in segs for w in s["words"]]
    assert len(segs) >= 12 and len(words) >= 60
    assert any(s.get("chars", 0) is None and s["words"] == [] for s in segs)          # align()'s failure branch (backtrack failed)
    assert any("score" in w and "start" in w for w in words)
    n_in = len([s for s in _segments() if len(s["audio"]) >= 4000])
    assert len(segs) > n_in                                                           # sentences were split
    assert all(s["start"] >= 0 for s in segs)


def test_world1_sharded_equals_transcribe_batch():
    """no process group: the sharded entry point is the single-process call, through pack -> unpack -> assemble"""
    for reuse in (False, True):          # False: every chunk, the rank's own too, is rebuilt from its record
        got = P.transcribe_batch_sharded(_HostBackend(), _segments(), batch_size=16, align_words=True, language="en", reuse_own=reuse)
        assert got == _single()
        asr = P.transcribe_batch_sharded(_HostBackend(), _segments(), batch_size=16, align_words=False, language="en", reuse_own=reuse)
        assert asr == _HostBackend().transcribe_batch(_segments(), batch_size=16, language="en")


def test_transcribe_and_align_sharded_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = _single()
    for _rank, got, n_coll, n_dec in outs:
        assert got["language"] == single["language"]
        assert len(got["segments"]) == len(single["segments"])
        for a, b in zip(got["segments"], single["segments"]):
            assert a == b, (a, b)
        assert n_coll == 1 and n_dec == 1            # ONE gather carries tokens, DTW spans and the aligned words


def test_record_roundtrip_of_an_align_result():
    be = _HostBackend()
    segs = _segments()
    res = be.transcribe_batch(segs, batch_size=16, language="en", return_chunks=True)
    groups = be._group_by_vad(res["segments"], segs)
    trace = []
    aligned = be.align_groups(groups, segs, "en", _trace=trace)
    for (vi, rel), a, tr in zip(groups, aligned, trace):
        row = np.zeros(P.REC_W, dtype=np.int32)
        P.pack_aligned(row, a, tr)
        back = P.assemble_aligned(rel, P.unpack_aligned(row), "en")
        assert back == a, vi
    row = np.zeros(P.REC_W, dtype=np.int32)
    row[P._O_ALIGN] = -1
    assert P.unpack_aligned(row) is None
    bad = {"segments": [{"start": 0.0005, "end": 1.0, "text": "x", "words": []}], "word_segments": []}
    try:
        P.pack_aligned(np.zeros(P.REC_W, dtype=np.int32), bad, [("ok", 0, [(0, 1)])])
        raise AssertionError("a time that is not whole milliseconds must not travel silently")
    except ValueError:
        pass
