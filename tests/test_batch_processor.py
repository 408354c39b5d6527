"""CPU: the long-form chunker against the reference's own (pure-Python) test cases
(/root/reference/tests/test_mlx_backend.py:310-391, which cannot be imported there because of
the mlx dependency) plus the window arithmetic of batch_processor.py:80-97."""
import numpy as np

from whisperx_mlx_amd.batch_processor import AudioChunk, BatchProcessor


def test_chunk_creation():
    p = BatchProcessor(batch_size=4, chunk_duration=30.0, overlap=0.5)
    segs = [{"start": 0.0, "end": 10.0}, {"start": 10.0, "end": 50.0}]
    audio = np.zeros(60 * 16000)
    chunks = p.create_chunks(audio, segs)
    assert len(chunks) >= 3
    for c in chunks:
        assert isinstance(c, AudioChunk) and isinstance(c.audio, np.ndarray)
        assert c.start_time >= 0 and c.end_time > c.start_time
    # 40 s segment: ceil(40 / 29.5) = 2 windows starting 29.5 s apart, the second clipped at the end
    assert [(c.start_time, c.end_time, c.segment_idx) for c in chunks] == [(0.0, 10.0, 0), (10.0, 40.0, 1), (39.5, 50.0, 1)]
    assert [len(c.audio) for c in chunks] == [160000, 480000, 168000]


def test_batch_creation():
    p = BatchProcessor(batch_size=3)
    chunks = [AudioChunk(np.zeros(1000), i, i + 1, 0) for i in range(5)]
    batches = p.create_batches(chunks)
    assert [len(b) for b in batches] == [3, 2]


def test_padding():
    p = BatchProcessor()
    chunks = [AudioChunk(np.ones(1000), 0, 1, 0), AudioChunk(np.ones(1500), 1, 2, 0), AudioChunk(np.ones(800), 2, 3, 0)]
    padded, lengths = p.pad_batch(chunks)
    assert padded.shape == (3, 1500) and lengths == [1000, 1500, 800]
    assert np.all(padded[0, :1000] == 1) and np.all(padded[0, 1000:] == 0)


def test_merge_results_and_overlap_heuristic():
    p = BatchProcessor()
    segs = [{"start": 0.0, "end": 10.0}, {"start": 10.0, "end": 50.0}, {"start": 50.0, "end": 51.0}]
    chunks = [AudioChunk(np.zeros(1), 0.0, 10.0, 0), AudioChunk(np.zeros(1), 39.5, 50.0, 1), AudioChunk(np.zeros(1), 10.0, 40.0, 1)]
    results = [{"text": " hello there "}, {"text": "a b c d e f g h i j"}, {"text": "one two three"}]
    out = p.merge_results(chunks, results, segs)
    assert out[0] == {"start": 0.0, "end": 10.0, "text": "hello there"}
    # chunks are re-ordered by start time; the later chunk loses its first 10 // 5 = 2 words
    assert out[1]["text"] == "one two three c d e f g h i j"
    assert out[2] == {"start": 50.0, "end": 51.0, "text": ""}


def test_scheduler_pass_sizes():
    """WhisperHipBackend's cut of a chunk list into passes (backend.pass_sizes): every row exactly once, never above
    rows_per_pass, whole rounds of full passes first and one balanced round for the remainder (no pass under 8 rows unless
    the remainder itself is smaller), and a modelled makespan (a pass costs a + b * rows, pass i on context i % lanes)
    never above that of the plain cut into full passes plus a remainder."""
    from whisperx_mlx_amd.backend import pass_sizes
    assert pass_sizes(81, 16, 3) == [16, 16, 16, 11, 11, 11]            # the reference run's 81 VAD windows
    assert pass_sizes(81, 16, 4) == [16, 16, 16, 16, 9, 8]
    assert pass_sizes(100, 16, 4) == [16, 16, 16, 16, 9, 9, 9, 9]
    assert pass_sizes(320, 16, 4) == [16] * 20 and pass_sizes(384, 16, 4) == [16] * 24
    assert pass_sizes(320, 48, 3) == [48] * 6 + [11, 11, 10]
    assert pass_sizes(5, 16, 3) == [5] and pass_sizes(0, 16, 3) == [0] and pass_sizes(17, 16, 4) == [9, 8]

    def makespan(sizes, lanes, a, b):
        return max(sum(a + b * r for r in sizes[k::lanes]) for k in range(min(lanes, len(sizes))))

    for n in list(range(1, 200)) + [320, 999, 1221]:
        for R in (4, 8, 16, 48):
            for lanes in (1, 2, 3, 4):
                s = pass_sizes(n, R, lanes)
                n_min = -(-n // R)
                assert sum(s) == n and max(s) <= R and min(s) >= 1 and n_min <= len(s) <= n_min + lanes - 1, (n, R, lanes, s)
                full = (n // (lanes * R)) * lanes
                assert s[:full] == [R] * full and len(s) - full <= lanes          # whole rounds first, one round for the rest
                if len(s) - full > -(-(n - full * R) // R):                      # the remainder was cut finer than necessary ...
                    assert min(s[full:]) >= min(8, R // 2)                         # ... but not into crumbs
                plain = [R] * (n // R) + ([n % R] if n % R else [])
                for a, b in ((5.0, 1.0), (1.0, 1.0), (0.0, 1.0)):
                    assert makespan(s, lanes, a, b) <= makespan(plain, lanes, a, b) + 1e-9, (n, R, lanes, a, b, s)


def test_plan_passes_merges_requests_into_wide_passes():
    """backend.plan_passes (the default scheduler, coalesce=None): every chunk exactly once; jobs worth three passes of
    more than 16 rows have their groups of 16 rows dealt evenly to three contexts (every pass decodes the same number of
    steps: equal rows end together), each context's share cut into passes of <= rows_cap rows of whole groups, issued
    round by round, the ragged group off the first pass; small jobs fall back to <= 16-row passes on up to four contexts."""
    from whisperx_mlx_amd.backend import pass_sizes, plan_passes
    assert plan_passes(320, 128) == ([112, 112, 96], 3)                   # the driver's bench job: 20 requests of 16 chunks
    assert plan_passes(320, 64) == ([64, 64, 48, 48, 48, 48], 3)
    assert plan_passes(768, 128)[0] == [128] * 6 and plan_passes(384, 64)[0] == [64] * 6
    assert plan_passes(400, 128) == ([80, 128, 128, 64], 3)               # context 0: 80 + 64, contexts 1 and 2: 128 each
    assert plan_passes(200, 128) == ([72, 64, 64], 3)
    assert plan_passes(100, 128) == ([36, 32, 32], 3)
    assert plan_passes(81, 128) == ([17, 32, 32], 3)                      # the reference run's 81 VAD windows
    assert plan_passes(60, 128) == ([15, 15, 15, 15], 4)                  # a 30-minute file in fixed windows: one round of <= 16-row passes on four contexts
    assert plan_passes(64, 128) == ([16] * 4, 4) and plan_passes(65, 128) == ([17, 32, 16], 3)
    assert plan_passes(60, 128, lanes_16=3) == ([28, 16, 16], 3)          # only three streams run side by side: the wide cut
    assert plan_passes(1221, 128)[0] == [101, 112, 112, 112, 112] + [96] * 7        # 10 h long-form
    assert plan_passes(5, 128) == ([5], 1) and plan_passes(17, 128) == ([9, 8], 2)
    assert plan_passes(48, 128) == (pass_sizes(48, 16, 3), 3)
    assert plan_passes(100, 16) == (pass_sizes(100, 16, 4), 4)            # contexts of 16 rows (coalesce=1): as before
    for cap in (64, 128):
        for n in range(0, 1400, 7):
            sizes, lanes = plan_passes(n, cap)
            assert sum(sizes) == n and max(sizes, default=0) <= cap and 1 <= lanes <= 4
            if 49 <= n <= 64:                                             # one round of <= 16-row passes on four contexts
                assert lanes == 4 and len(sizes) == 4 and max(sizes) <= 16 and max(sizes) - min(sizes) <= 1
            if n > 64:
                assert lanes == 3 and len(sizes) >= 3
                assert all(v % 16 == 0 for v in sizes[1:])                # only the first pass may hold the ragged row group
                per_ctx = [sum(sizes[k::3]) for k in range(3)]            # pass i runs on context i % 3
                assert max(per_ctx) - min(per_ctx) <= 16 + 15             # whole groups dealt evenly (+ the ragged one)
    sizes, lanes = plan_passes(200, 64, lanes_16=2, lanes_wide=2)          # fewer streams run side by side
    assert lanes == 2 and sum(sizes) == 200
