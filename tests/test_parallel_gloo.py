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


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dur = [30.0, 12.0, 29.0, 5.0, 17.0, 30.0, 1.0]
        mine = P.shard_indices(dur, rank, world)
        results = [{"tokens": [50365, 100 + i, 50257 - 1], "sum_logprob": -float(i), "no_speech_prob": i / 10.0,
                    "word_spans": [(1, i, i + 20)]} for i in mine]
        got = P.gather_records(P.pack_records(results, mine))
        q.put((rank, [(g["chunk_id"], g["tokens"][1], g["sum_logprob"], g["word_spans"]) for g in got]))
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
    for _rank, got in outs:
        assert got == expect
