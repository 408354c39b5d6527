"""-m gpu: determinism of the decode loop under repetition and with three passes in flight (the in-launch merge of the
cross-attention key splits hands partial results between blocks through tagged granules; hipGraphs are replayed from
several launcher threads).  The long version is tools/stress_decode.py."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import gpu_util as G                       # noqa: E402
from whisperx_mlx_amd import engine as E               # noqa: E402
from whisperx_mlx_amd.tokenizer import get_tokenizer   # noqa: E402


def test_repeated_and_concurrent_decodes_are_identical():
    from whisperx_mlx_amd import weights
    from whisperx_mlx_amd.engine import WhisperHipEngine
    ck = weights.random_checkpoint(G.TEST_DIMS, seed=0, std=0.2, emb_std=0.1)
    # three CONTEXTS (G.tiny_engine is cached: it would hand out one context three times, and a context is
    # single-threaded by contract -- the library refuses concurrent entry, see test_context_refuses_concurrent_entry)
    engs = [WhisperHipEngine(G.TEST_DIMS, weights.pack(ck, G.TEST_DIMS, "cuda"), max_batch=8) for _ in range(3)]
    dims = engs[0].dims
    tok = get_tokenizer(dims.n_vocab)
    g = torch.Generator().manual_seed(5)
    mel = (torch.randn(8, 3000, dims.n_mels, generator=g) * 0.5).half().cuda()
    ref = {}
    for it in range(24):
        e = engs[it % 2]
        split = (4, 2, 1)[it % 3]
        out = e.decode(e.encode(mel), tok, tok.sot_sequence(), rules=0, forced_len=32, cross_split=split)
        e.check_status()
        t = out.tokens.cpu().numpy().copy()
        assert np.array_equal(ref.setdefault(split, t), t), (it, split)
    # every context captures its hipGraphs for the options it will run with BEFORE the threads start (one capture at a
    # time: a capture that races with another thread's allocator traffic can be rejected by the runtime)
    for e in engs:
        e.decode(e.encode(mel), tok, tok.sot_sequence(), rules=0, forced_len=32, cross_split=2, fc2_tile_n=16)
        e.check_status()
    results = [[] for _ in engs]
    errors = []

    def worker(k):
        try:
            torch.cuda.set_device(0)
            e = engs[k]
            with torch.cuda.stream(e.stream):
                for _ in range(4):
                    out = e.decode(e.encode(mel), tok, tok.sot_sequence(), rules=0, forced_len=32, cross_split=2, fc2_tile_n=16)
                    e.check_status()
                    results[k].append(out.tokens.cpu().numpy().copy())
        except Exception as ex:       # noqa: BLE001
            errors.append(ex)

    th = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    for rs in results:
        assert len(rs) == 4
        for t in rs:
            assert np.array_equal(t, ref[2])


def test_context_refuses_concurrent_entry():
    """One context = one launcher thread: a second thread entering the same context while a call is in progress gets an
    error instead of silently corrupting the workspace."""
    from whisperx_mlx_amd._lib import WxError
    eng = G.tiny_engine(max_batch=8)[0]
    tok = get_tokenizer(eng.dims.n_vocab)
    mel = (torch.randn(8, 3000, eng.dims.n_mels) * 0.5).half().cuda()
    enc = eng.encode(mel)
    torch.cuda.synchronize()
    outcomes = []

    def worker():
        torch.cuda.set_device(0)
        try:
            for _ in range(6):
                eng.decode(enc, tok, tok.sot_sequence(), rules=0, forced_len=48, use_graph=False)
            outcomes.append("ok")
        except WxError as ex:
            outcomes.append("refused" if "another thread" in str(ex) else repr(ex))

    th = [threading.Thread(target=worker) for _ in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    assert all(o in ("ok", "refused") for o in outcomes), outcomes
    assert "refused" in outcomes            # the three threads did collide
    # the context is still usable afterwards
    a = eng.decode(enc, tok, tok.sot_sequence(), rules=0, forced_len=16).tokens.cpu().numpy().copy()
    b = eng.decode(enc, tok, tok.sot_sequence(), rules=0, forced_len=16).tokens.cpu().numpy().copy()
    eng.check_status()
    assert np.array_equal(a, b)


def test_fused_launch_makes_progress_without_its_producers():
    """The forward-progress guarantee of the fused decode launch, exercised on purpose (wx_test_fused_selfq: the attention
    blocks poll a granule buffer nobody publishes to).  Every one of them gives up polling after ~100 us and computes its
    query itself with the producer role's own code: the launch drains at once, its output is BIT-IDENTICAL to the two
    launches it stands for, nothing is flagged or poisoned, and the context decodes exactly what it decoded before.
    Both the tiny model and large-v3 width (d = 1280: 8 GEMV tiles per head), at 16 rows and -- ADVICE r03 -- at 40 and 128
    rows: a block of row b then works for row group b >> 4 and selects row b & 15 of it, which 16-row launches never visit."""
    import ctypes as C
    import time
    from whisperx_mlx_amd import _lib, weights
    from whisperx_mlx_amd.engine import WhisperHipEngine
    from whisperx_mlx_amd.tokenizer import get_tokenizer
    wide = weights.ModelDimensions(128, 1500, 1280, 20, 1, 51866, 448, 1280, 20, 2)
    ckw = weights.random_checkpoint(wide, seed=5, std=0.02)
    engw = WhisperHipEngine(wide, weights.pack(ckw, wide, "cuda"), max_batch=16)
    # the same width with int8 decoder weights (layer 0, the one the hook launches, is int8: the Q8 instance of the kernel)
    engq = WhisperHipEngine(wide, weights.quantize_packed_decoder(weights.pack(ckw, wide, "cuda"), wide), max_batch=16)
    engw128 = WhisperHipEngine(wide, engw.packed, max_batch=128)
    engq128 = WhisperHipEngine(wide, engq.packed, max_batch=48)
    for eng, dims, B in ((G.tiny_engine()[0], G.TEST_DIMS, 4), (engw, wide, 16), (engq, wide, 16), (engw128, wide, 40),
                         (engw128, wide, 128), (engq128, wide, 37)):
        tok = get_tokenizer(dims.n_vocab)
        enc = eng.encode((torch.randn(B, 3000, dims.n_mels, generator=torch.Generator().manual_seed(3)) * 0.5).half().cuda())
        kw = dict(rules=E.RULES_LIGHTNING, suppress_ids=tok.suppress_tokens(), sample_len=12)
        before = eng.decode(enc, tok, tok.sot_sequence(), **kw).tokens.cpu().numpy().copy()
        eng.check_status()
        assert eng.decode_stats()["selfq"] == 0               # the normal run never needed the path
        fused = torch.zeros(B, dims.n_text_state, dtype=torch.float16, device="cuda")
        ref = torch.ones(B, dims.n_text_state, dtype=torch.float16, device="cuda")
        n = C.c_int(-1)
        t0 = time.perf_counter()
        rc = _lib.lib().wx_test_fused_selfq(eng.ctx, B, _lib.ptr(fused), _lib.ptr(ref), C.byref(n), eng._s)
        dt = time.perf_counter() - t0
        _lib.check(eng.ctx, rc, "wx_test_fused_selfq")
        assert dt < 2.0
        assert n.value == B * dims.n_text_head                 # every attention block took the path
        assert torch.isfinite(fused.float()).all() and float(ref.float().abs().max()) > 0
        assert torch.equal(fused, ref)                         # the same bits as LN + GEMV, then cross attention
        eng.check_status()                                     # nothing flagged
        after = eng.decode(enc, tok, tok.sot_sequence(), **kw).tokens.cpu().numpy()
        eng.check_status()
        assert np.array_equal(before, after)
    for e in (engw128, engq128, engw, engq):
        e.close()


def test_two_hundred_passes_at_large_v3_width_four_in_flight():
    """tools/stress_fused.py as a test: 200 passes of 16 rows at large-v3 width (d = 1280, 20 heads, 2 + 2
    layers) on FOUR contexts in flight -- the configuration in which consumers of one fused launch can hold the wave slots
    another launch's producers need.  Every repeat of a chunk decodes to the same tokens and log-probability, nothing is
    flagged, no key-split give-up, and the blocks that computed their query themselves are counted (any number is fine:
    they produce the same bits)."""
    import warnings
    from whisperx_mlx_amd import backend as BK
    from whisperx_mlx_amd.synth import speechlike_audio
    BK.W.MODEL_DIMS.setdefault("stress-wide", BK.W.ModelDimensions(128, 1500, 1280, 20, 2, 51866, 448, 1280, 20, 2))
    be = BK.WhisperHipBackend("stress-wide", random_init=True, seed=2, max_batch=16, coalesce=1, init_std=0.03, init_emb_std=0.03,
                              passes_in_flight=4)
    dev = torch.from_numpy(np.stack([speechlike_audio(30.0, seed=800 + i) for i in range(32)])).cuda()
    chunks = [dev[i % 32] for i in range(200 * 16)]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = be._decode_chunks(chunks, "en", "transcribe", False, forced_len=48, rows_per_pass=16, passes_in_flight=4)
    assert be.last_plan["passes_in_flight"] == 4 and len(be.last_plan["rows"]) == 200
    assert not [x for x in w if "gave up" in str(x.message)] and be.split_giveups == 0
    for i, r in enumerate(out):
        assert r["tokens"] == out[i % 32]["tokens"] and r["sum_logprob"] == out[i % 32]["sum_logprob"], i
    assert len(out[0]["tokens"]) == 48
    for e in be.engines:
        e.check_status()
    print(f"self-computed queries in 200 passes: {be.selfq_blocks}")


def test_launch_timer_counts_every_fused_launch_and_changes_nothing():
    """wx_decode_opts.profile_launches (bench.py's live roofline figure): every fused launch of the decode is timed once,
    the average is a plausible duration, and tokens / log-probabilities are those of the untimed decode"""
    from whisperx_mlx_amd import weights
    from whisperx_mlx_amd.engine import WhisperHipEngine
    wide = weights.ModelDimensions(128, 1500, 1280, 20, 1, 51866, 448, 1280, 20, 2)
    eng = WhisperHipEngine(wide, weights.pack(weights.random_checkpoint(wide, seed=5, std=0.03, emb_std=0.03), wide, "cuda"), max_batch=16)
    tok = get_tokenizer(wide.n_vocab)
    enc = eng.encode((torch.randn(16, 3000, wide.n_mels, generator=torch.Generator().manual_seed(3)) * 0.5).half().cuda())
    kw = dict(rules=0, forced_len=20)
    a = eng.decode(enc, tok, tok.sot_sequence(), **kw)
    ta, la = a.tokens.cpu().numpy().copy(), a.sum_logprob.cpu().numpy().copy()
    assert eng.launch_profile() == (0.0, 0)
    b = eng.decode(enc, tok, tok.sot_sequence(), profile_launches=True, **kw)
    assert np.array_equal(b.tokens.cpu().numpy(), ta) and np.array_equal(b.sum_logprob.cpu().numpy(), la)
    avg, n = eng.launch_profile()
    assert n == (len(tok.sot_sequence()) + 20 - 1) * wide.n_text_layer        # one per position and layer
    assert 5.0 < avg < 200.0, avg                                             # ~25 us at 16 rows
    assert eng.launch_profile() == (0.0, 0)                                   # read and clear
    eng.close()


def test_logmel_and_encoder_are_bit_stable_beside_another_contexts_gemms():
    """Round 2 finding: packed-fp32 VALU results change in lanes 48-63 while another wave of the SIMD
    issues MFMAs, so the log-mel DFT of one context came out different whenever it shared CUs with the 128^2 GEMM
    tiles of another context (26 of 40 repeats).  The library is built without packed-fp32 ops; every stage of one
    context must return the same bits whatever the neighbours run."""
    from tests.synth import speechlike_audio
    from whisperx_mlx_amd import weights
    from whisperx_mlx_amd.engine import WhisperHipEngine
    ck = weights.random_checkpoint(G.TEST_DIMS, seed=0, std=0.2, emb_std=0.1)
    packed = weights.pack(ck, G.TEST_DIMS, "cuda")
    e0, e1 = (WhisperHipEngine(G.TEST_DIMS, packed, max_batch=8) for _ in range(2))
    tok = get_tokenizer(e0.dims.n_vocab)
    pcm = torch.from_numpy(np.stack([speechlike_audio(30.0, seed=90 + i) for i in range(5)])).cuda()
    nv = torch.full((5,), 480000, dtype=torch.int32, device="cuda")

    def stages(e):
        mel, mel32 = e.logmel(pcm, nv, want_f32=True)
        enc = e.encode(mel)
        out = e.decode(enc, tok, tok.sot_sequence(), rules=0, forced_len=24, cross_split=2)
        e.check_status()
        return mel32.clone(), enc.clone(), out.tokens.clone(), out.sum_logprob.clone()

    with torch.cuda.stream(e0.stream):
        ref = stages(e0)
    with torch.cuda.stream(e1.stream):
        mel1 = e1.logmel(pcm, nv).clone()
        e1.encode(mel1)
    torch.cuda.synchronize()
    stop, errors = [False], []

    def neighbour():
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(e1.stream):
                while not stop[0]:
                    e1.encode(mel1)                 # FC2 / out-proj of this shape run on the 128^2 MFMA GEMM
                    e1.stream.synchronize()
        except Exception as ex:       # noqa: BLE001
            errors.append(ex)

    th = threading.Thread(target=neighbour)
    th.start()
    differing = {"logmel": 0, "encoder": 0, "tokens": 0, "sum_logprob": 0}
    try:
        with torch.cuda.stream(e0.stream):
            for _ in range(20):
                got = stages(e0)
                for name, a, b in zip(differing, got, ref):
                    differing[name] += int(not torch.equal(a, b))
    finally:
        stop[0] = True
        th.join()
    assert not errors, errors
    assert differing == {"logmel": 0, "encoder": 0, "tokens": 0, "sum_logprob": 0}, differing
