"""CPU: the N>1 path (chunk sharding + the single gather) with world_size 2 on gloo."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from whisperx_mlx_amd import parallel as P


def test_shard_indices_partition_and_balance():
    rng = np.random.default_rng(0)
    dur = rng.uniform(1, 30, 61).tolist()
    for world in (1, 2, 4, 8):
        shards = [P.shard_indices(dur, r, world) for r in range(world)]
        allidx = sorted(i for s in shards for i in s)
        assert allidx == list(range(61))
        loads = [sum(dur[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= 30.0
    assert P.shard_indices([], 0, 2) == []


def test_pack_unpack_roundtrip():
    res = [{"tokens": [50365, 1, 2, 3, 50465], "sum_logprob": -3.25, "no_speech_prob": 0.125,
            "word_spans": [(2, 0, 480), (4, 480, 1000)]},
           {"tokens": [], "sum_logprob": 0.0, "no_speech_prob": 1.0, "word_spans": []},
           {"tokens": list(range(300)), "sum_logprob": -1e3, "no_speech_prob": 0.0, "word_spans": []}]
    out = P.unpack_records(P.pack_records(res, [7, 3, 11]))
    assert [o["chunk_id"] for o in out] == [7, 3, 11]
    assert out[0]["tokens"] == res[0]["tokens"] and out[0]["word_spans"] == res[0]["word_spans"]
    assert out[0]["sum_logprob"] == -3.25 and out[1]["no_speech_prob"] == 1.0
    assert out[2]["tokens"] == list(range(224))          # truncated to sample_len


def _count_collectives():
    """wraps the collectives torch.distributed offers for tensors and counts the calls (the north star allows one)"""
    calls = {"n": 0}
    for name in ("all_gather_into_tensor", "all_gather", "all_reduce", "broadcast", "gather", "all_to_all"):
        fn = getattr(dist, name)

        def wrapped(*a, _fn=fn, **k):
            calls["n"] += 1
            return _fn(*a, **k)
        setattr(dist, name, wrapped)
    return calls


class _FakeBackend:
    """host-side stand-in for WhisperHipBackend._decode_chunks: tokens and words derived from the chunk's content, so
    that every rank's output can be checked against a single-process run"""
    is_multilingual = True

    def __init__(self):
        self.detect_calls = 0

    def detect_language(self, chunk):
        self.detect_calls += 1
        return "de" if float(chunk[0]) > 0 else "en"

    def _decode_chunks(self, chunks, language, task, word_timestamps, **kw):
        out = []
        for c in chunks:
            n = 3 + int(abs(float(c[0])) * 10) % 5
            toks = [50365] + [1000 + int(len(c) % 977) + k for k in range(n)] + [50365 + len(c) // 320]
            words = [{"word": f"w{k}", "start": 0.02 * k, "end": 0.02 * k + 0.5, "probability": 1.0} for k in range(2)]
            out.append({"tokens": toks, "text": "x", "avg_logprob": -0.5, "sum_logprob": -0.5 * (len(toks) + 1),
                        "no_speech_prob": 0.25, "language": language, "words": words, "word_token_counts": [2, n - 2]})
        return out


def _chunks_for_sharded():
    rng = np.random.default_rng(5)
    return [np.full(int(n), 0.1 * (i + 1), dtype=np.float32) for i, n in enumerate(rng.integers(16000, 480000, 9))]


def _sharded_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = _count_collectives()
        be = _FakeBackend()
        got = P.transcribe_sharded(be, _chunks_for_sharded(), language=None, word_timestamps="dtw")
        q.put((rank, got, calls["n"], be.detect_calls))
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dur = [30.0, 12.0, 29.0, 5.0, 17.0, 30.0, 1.0]
        mine = P.shard_indices(dur, rank, world)
        results = [{"tokens": [50365, 100 + i, 50257 - 1], "sum_logprob": -float(i), "no_speech_prob": i / 10.0,
                    "word_spans": [(1, i, i + 20)]} for i in mine]
        counts = [len(P.shard_indices(dur, r, world)) for r in range(world)]       # known on every rank: ONE collective
        calls = _count_collectives()
        got = P.gather_records(P.pack_records(results, mine), counts=counts)
        q.put((rank, [(g["chunk_id"], g["tokens"][1], g["sum_logprob"], g["word_spans"]) for g in got], calls["n"]))
    finally:
        dist.destroy_process_group()


def test_gather_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [(i, 100 + i, -float(i), [(1, i, i + 20)]) for i in range(7)]
    for _rank, got, n_coll in outs:
        assert got == expect
        assert n_coll == 1            # north star: a single gather at the end, no size exchange


def test_transcribe_sharded_world2_gloo():
    """the N > 1 product path with a fake backend: every rank ends with the single-process result (token ids,
    log-probabilities, word spans with advancing token positions, language detected once per rank on the SAME chunk),
    through exactly one collective"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    chunks = _chunks_for_sharded()
    single = P.transcribe_sharded(_FakeBackend(), chunks, language=None, word_timestamps="dtw")     # no process group: world 1
    assert [r["chunk_id"] for r in single] == list(range(9))
    for r in single:
        ends = [w[0] for w in r["word_spans"]]
        assert ends == sorted(ends) and ends[0] == 2 and ends[-1] == len(r["tokens"]) - 2      # positions advance
        assert r["word_spans"][1][1:] == (20, 520)
    for _rank, got, n_coll, n_detect in outs:
        assert got == single
        assert n_coll == 1 and n_detect == 1
