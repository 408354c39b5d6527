"""`backend="hip"`: the WhisperBackend the reference's pipeline calls
(/root/reference/whisperx/backends/base.py:8-57; template
/root/reference/whisperx/backends/mlx_lightning.py:20-406), running the whole hot path
on one MI355X through libwxhip.so.

    model = load_model("large-v3", device="cuda", backend="hip", download_root="/ckpts")
    result = model.transcribe(audio, batch_size=16)

Differences from the lightning backend that are deliberate: chunks are decoded B at a
time in one device-resident greedy loop (the reference loops segments sequentially,
mlx_lightning.py:82-119), and the log-mel is computed per 30 s chunk on the GPU
(mlx_whisper_optimized_final.py:428-434) instead of once per file on the host.
"""
import ctypes as C
import os
import subprocess
import threading
import warnings
import zlib
from abc import ABC, abstractmethod
from typing import Any, Dict, List, Optional, Union

import numpy as np
import torch

from . import weights as W
from .audio import N_SAMPLES, SAMPLE_RATE, TOKENS_PER_SECOND
from . import _lib
from ._lib import WxError
from .engine import RULES_LIGHTNING, WhisperHipEngine
from .tokenizer import LANGUAGES, get_tokenizer


def _load_pcm_without_ffmpeg(file: str, sr: int) -> np.ndarray:
    """.npy (mono float array at `sr`) and RIFF/WAVE PCM (8/16/32-bit int, mono or multi-channel: channels are
    averaged, other rates are resampled with a polyphase filter) for boxes without the ffmpeg binary."""
    if file.endswith(".npy"):
        return np.load(file).astype(np.float32).reshape(-1)
    import wave
    with wave.open(file, "rb") as w:
        n_ch, width, rate, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if width == 2:
        x = np.frombuffer(raw, np.int16).astype(np.float32) / 32768.0
    elif width == 4:
        x = np.frombuffer(raw, np.int32).astype(np.float32) / 2147483648.0
    elif width == 1:
        x = (np.frombuffer(raw, np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise RuntimeError(f"Failed to load audio: unsupported sample width {width} in {file}")
    if n_ch > 1:
        x = x.reshape(-1, n_ch).mean(axis=1)
    if rate != sr:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(rate, sr)
        x = resample_poly(x, sr // g, rate // g).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32)


def load_audio(file: str, sr: int = SAMPLE_RATE) -> np.ndarray:
    """whisperx/audio.py:25-65: decode with the ffmpeg CLI to mono f32 at `sr`.  Without an ffmpeg binary on
    the PATH, .wav (PCM) and .npy inputs are read directly."""
    try:
        cmd = ["ffmpeg", "-nostdin", "-threads", "0", "-i", file, "-f", "s16le", "-ac", "1", "-acodec", "pcm_s16le",
               "-ar", str(sr), "-"]
        out = subprocess.run(cmd, capture_output=True, check=True).stdout
    except subprocess.CalledProcessError as e:
        raise RuntimeError(f"Failed to load audio: {e.stderr.decode()}") from e
    except FileNotFoundError:
        return _load_pcm_without_ffmpeg(file, sr)
    return np.frombuffer(out, np.int16).flatten().astype(np.float32) / 32768.0


class WhisperBackend(ABC):
    """Same abstract surface as whisperx/backends/base.py:8-57."""

    @abstractmethod
    def __init__(self, model: str, device: str, device_index: int = 0, compute_type: str = "float16",
                 download_root: Optional[str] = None, local_files_only: bool = False, threads: int = 4, **kwargs):
        pass

    @abstractmethod
    def transcribe(self, audio, batch_size=None, num_workers=0, language=None, task=None, chunk_size=30,
                   print_progress=False, combined_progress=False, verbose=False, **kwargs):
        pass

    @abstractmethod
    def detect_language(self, audio: np.ndarray) -> str:
        pass

    @property
    @abstractmethod
    def supported_languages(self) -> List[str]:
        pass

    @property
    @abstractmethod
    def is_multilingual(self) -> bool:
        pass


_engine_cache: Dict[str, Any] = {}       # like mlx_lightning.py:17 (one model per process)


from .scheduler import DEFAULT_ROWS, MAX_ROWS, pass_sizes, plan_job, plan_passes      # noqa: E402,F401  (re-exported: the round-2/3 names)
from . import dtw_words as DW      # noqa: E402


from .contexts import PassSlot as _PassSlot, first_context, get_contexts, is_oom as _is_oom, new_context as _new_context      # noqa: E402,F401
from .lanes import Lane, PassJob      # noqa: E402


class WhisperHipBackend(WhisperBackend):
    def __init__(self, model: str, device: str = "cuda", device_index: int = 0, compute_type: str = "float16",
                 download_root: Optional[str] = None, local_files_only: bool = False, threads: int = 4,
                 max_batch: int = 16, random_init: bool = False, seed: int = 0, coalesce: Optional[int] = None, **kwargs):
        if compute_type not in ("float16", "fp16", "default", "int8"):
            raise ValueError(f"backend 'hip' computes in float16, optionally with int8 decoder weights (got compute_type={compute_type!r})")
        self.model_name = model
        # "int8": the decode GEMV weights are stored as int8 + row scales and dequantised in registers; activations,
        # accumulation, the encoder and the cross K/V projection stay as in float16 (weights.quantize_packed_decoder)
        self.compute_type = "int8" if compute_type == "int8" else "float16"
        self.device_index = device_index
        name = W.resolve_model_name(model)
        # Several requests of `max_batch` chunks may share one pass of the hot path (rows are independent; the decoder
        # weights are then streamed once per pass instead of once per request, and the cross-attention launch is wider):
        # the contexts take max_batch * coalesce rows.  coalesce=None (default): contexts of 128 rows and the scheduler
        # decides per job (plan_passes); coalesce=1: every pass is one request of max_batch chunks.
        self.auto_rows = coalesce is None
        rows_cap = max(16, min(int(kwargs.get("max_rows") or DEFAULT_ROWS), MAX_ROWS))     # max_rows=...: smaller contexts (less memory)
        max_rows = rows_cap if self.auto_rows else min(max(1, max_batch) * max(1, int(coalesce)), MAX_ROWS)
        key = (f"{model}|{device_index}|{download_root}|{random_init}|{seed}|{max_rows}|{self.compute_type}|"
               f"{kwargs.get('init_std')}|{kwargs.get('init_emb_std')}")
        if key not in _engine_cache:
            ckpt_dir = None
            for cand in (model, os.path.join(download_root or "", model), os.path.join(download_root or "", name)):
                if cand and os.path.isdir(cand) and os.path.exists(os.path.join(cand, "config.json")):
                    ckpt_dir = cand
                    break
            extra = {}
            dev = torch.device("cuda", device_index)
            if ckpt_dir:
                dims, sd, extra = W.load_checkpoint_dir(ckpt_dir)
            elif random_init:
                dims = W.MODEL_DIMS[name]
                sd = W.random_checkpoint(dims, seed=seed, device=dev, std=float(kwargs.get("init_std", 0.02)),
                                         emb_std=kwargs.get("init_emb_std"))
            else:
                raise FileNotFoundError(
                    f"no local checkpoint for {model!r} (looked in {download_root!r}); the GPU box has no network. "
                    "Pass download_root=<dir with config.json + safetensors> or random_init=True for a throughput run")
            heads = extra.get("alignment_heads") or W.default_alignment_heads(name, dims)
            packed = W.pack(sd, dims, dev)
            if self.compute_type == "int8":
                packed = W.quantize_packed_decoder(packed, dims)
            eng = first_context(dims, packed, max_rows, device_index, heads)
            _engine_cache[key] = ([eng], dims, ckpt_dir, extra)
        self.engines, self.dims, self.ckpt_dir, self.extra = _engine_cache[key]
        self.engine = self.engines[0]
        max_rows = self.engine.max_batch          # (smaller than asked for when the workspace did not fit)
        self.max_batch = max_batch
        # scheduler (see _decode_chunks): rows per pass of the hot path and passes in flight (engine contexts)
        self.rows_per_pass = max_rows
        self.coalesce = max(1, max_rows // max(1, max_batch)) if self.auto_rows else max(1, int(coalesce))
        self.cross_split = int(kwargs.get("cross_split", 0))       # 0: the default (2)
        self.fc2_tile_n = kwargs.get("fc2_tile_n")                 # None: by the number of passes in flight
        self.step_variant = int(kwargs.get("step_variant", 0))     # 0: fused launches (csrc/declayer.hip); 1: a kernel per stage
        # passes in flight: one engine context (stream + launcher thread) each.  Given explicitly, or settled on first use:
        # as many as there are engine streams that really run side by side, four at most (_default_lanes)
        self._lanes_req = int(kwargs.get("passes_in_flight") or 0)
        if self._lanes_req > 4:
            warnings.warn(f"passes_in_flight={self._lanes_req}: beyond four streams the GPU's hardware queues share dispatch pipes and "
                          "every extra pass in flight costs throughput (5 passes: 670x against 2 330x with 4 on large-v3)")
        self.passes_in_flight = self._lanes_req or (3 if max_rows <= 16 else 2)      # until _default_lanes() has asked the hardware
        self.stage_ms = None        # set to {} to collect per-stage GPU times (HIP events on the passes' own streams)
        self.profile_launches = bool(kwargs.get("profile_launches", False))   # bench.py: the fused launches time themselves (engine.launch_profile)
        self.last_plan = None       # how the last scheduler run cut its job: rows per pass, launch shape, passes in flight
        self.selfq_blocks = 0       # attention blocks of fused decode launches that computed their query themselves (engine.decode_stats)
        self.split_giveups = 0      # scheduler runs decoded again because a key-split merge wait expired (step_variant 1 only)
        self.dtw_variant = kwargs.get("dtw_variant", "upstream")   # "inrepo": mlx_whisper_optimized_final.py:128-253
        self.temperature = kwargs.get("temperature", 0.0)      # greedy only (mlx_lightning.py:77)
        self.tokenizer = get_tokenizer(self.dims.n_vocab, model_dir=self.ckpt_dir)
        self.suppress = self.tokenizer.suppress_tokens(self.extra.get("suppress_tokens"))
        self.rules = kwargs.get("rules", RULES_LIGHTNING)
        # one scheduler run at a time over these engine contexts (a context is single-threaded; backend objects built with
        # the same arguments share the contexts, hence the lock lives with them)
        if getattr(self.engine, "call_lock", None) is None:
            self.engine.call_lock = threading.RLock()
        self._call_lock = self.engine.call_lock
        self.align_model_cache = {}
        self.align_model_dir = kwargs.get("align_model_dir", download_root)

    # ------------------------------------------------------------------ core
    def _get_engines(self, n, rows=None):
        """the first `n` engine contexts, sized by the job (contexts.get_contexts: the memory policy)"""
        return get_contexts(self, n, rows)

    def _default_lanes(self, rows_per_pass: Optional[int] = None, need: Optional[int] = None) -> int:
        """Passes in flight when the caller does not say: every pass needs an engine stream with a hardware queue of its
        own.  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES queues (4 unless the variable is set before the GPU is
        first touched) in creation order, and streams that share a queue run one after the other: a pass on a stream
        that shares its queue makes the job SLOWER (four 16-row passes on colliding streams 1 880x against 2 190x with
        three).  Which streams collide cannot be read back, so the streams are asked once (wx_streams_overlap, ~3 ms a
        question): a context whose stream does not run beside the ones already chosen gets another stream from torch's
        pool (hipGraphs are launched on whatever stream the context has), and when none does, fewer passes are kept in
        flight.  Four at most: beyond four the hardware's queues share dispatch pipes and every extra pass costs
        (tools/ab_rows_lanes.py).  Passes above 16 rows (coalesced requests): one fewer.  `need`: the passes the job has --
        a short job neither creates nor examines contexts it cannot use."""
        R = rows_per_pass or self.rows_per_pass
        if self._lanes_req:
            return self._lanes_req
        cap = 4 if R <= 16 else 3
        want = max(1, min(cap, need or cap))           # a short job does not need (or create) every context
        found = getattr(self.engine, "side_by_side", 1)      # streams found to run side by side so far
        tested = getattr(self.engine, "side_by_side_tested", 1)   # contexts examined so far
        if want > tested and found == tested:
            L = _lib.lib()

            def side_by_side(streams):
                arr = (C.c_void_p * len(streams))(*[C.c_void_p(st.cuda_stream) for st in streams])
                f = C.c_float(0.0)
                return L.wx_streams_overlap(self.device_index, arr, len(streams), 300, C.byref(f)) == 0 and f.value < 1.5

            engs = self._get_engines(want, rows=R)
            chosen = [e.stream for e in engs[:found]]
            for e in engs[tested:]:
                tested += 1
                cands = [e.stream] + ([] if e.warm else [torch.cuda.Stream(device=e.device) for _ in range(12)])
                for st in cands:
                    if all(st.cuda_stream != c.cuda_stream for c in chosen) and side_by_side(chosen + [st]):
                        e.stream = st
                        chosen.append(st)
                        break
                else:
                    warnings.warn(f"only {len(chosen)} engine streams run side by side on this process's hardware queues "
                                  f"(GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'unset')}, possibly set after the GPU "
                                  f"was initialised): {len(chosen)} passes in flight instead of {want}")
                    break
            found = len(chosen)
            self.engine.side_by_side, self.engine.side_by_side_tested = found, tested
        n = min(want, found)
        if R == self.rows_per_pass and (not need or need >= cap):      # for reports: the default this backend settled on
            self.passes_in_flight = n
        return n

    def _slots(self, eng):
        if getattr(eng, "pass_slots", None) is None:
            eng.pass_slots = [_PassSlot(eng.max_batch, self.dims, eng.device) for _ in range(2)]
        return eng.pass_slots

    def _enqueue_pass(self, eng, slot, batch, prompt, dtw, forced_len, cross_split, fc2_tile_n, forced_lens=None, launch_rows=None,
                      steps_ahead=0):
        """One pass of the hot path over <= rows_per_pass chunks, enqueued on the engine's stream with no host
        synchronisation in this function (a free-running decode polls its all-done flag from inside wx_decode_greedy):
        PCM staging -> log-mel -> encoder -> greedy decode -> alignment matrix + DTW -> results into the slot's pinned
        host buffers.  `slot.event` says when they have landed."""
        n = len(batch)
        st = eng.stream
        batch = [c if torch.is_tensor(c) else np.asarray(c, dtype=np.float32).reshape(-1) for c in batch]
        on_dev = [torch.is_tensor(c) and c.is_cuda for c in batch]
        host = slot.host(n, want_pcm=not all(on_dev))
        lens = [min(int(c.shape[0]), N_SAMPLES) for c in batch]
        with torch.cuda.stream(st):
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if self.stage_ms is not None else None
            mark = (lambda k: marks[k].record(st)) if marks else (lambda k: None)
            pcm = torch.zeros(n, N_SAMPLES, dtype=torch.float32, device=eng.device)
            if not all(on_dev):
                stage = host["pcm"][:n]
                stage.zero_()
                for i, c in enumerate(batch):
                    if not on_dev[i]:
                        stage[i, : lens[i]] = (c if torch.is_tensor(c) else torch.from_numpy(c))[: lens[i]]
                pcm.copy_(stage, non_blocking=True)
            for i, c in enumerate(batch):
                if on_dev[i]:
                    pcm[i, : lens[i]].copy_(c[: lens[i]], non_blocking=True)
            host["nv"][:n] = torch.tensor(lens, dtype=torch.int32)
            nv = host["nv"][:n].to(eng.device, non_blocking=True)
            mark(0)
            mel = eng.logmel(pcm, nv)
            mark(1)
            enc = eng.encode(mel)
            mark(2)
            fl = None
            if forced_lens is not None:           # bench workload: per-row lengths (the caller's forced_len is their bound)
                host["flen"][:n] = torch.tensor([min(int(v), forced_len) for v in forced_lens], dtype=torch.int32)
                fl = slot.flen_dev
                fl[:n].copy_(host["flen"][:n], non_blocking=True)
            dec = eng.decode(enc, self.tokenizer, prompt, rules=self.rules, suppress_ids=self.suppress, forced_lens=fl,
                             capture_qk=bool(dtw), forced_len=forced_len, cross_split=cross_split, fc2_tile_n=fc2_tile_n,
                             step_variant=1 if cross_split != 2 else self.step_variant, rows=launch_rows,
                             profile_launches=self.profile_launches, max_steps_ahead=steps_ahead)
            mark(3)
            slot.n, slot.n_prompt, slot.n_sampled, slot.lens = n, dec.n_prompt, dec.n_sampled, lens
            host["tokens"][:n].copy_(dec.tokens, non_blocking=True)
            host["sum_lp"][:n].copy_(dec.sum_logprob, non_blocking=True)
            host["nsp"][:n].copy_(dec.no_speech_prob, non_blocking=True)
            if dtw:
                # published find_alignment crops the attention to the frames that carry audio (num_frames // 2)
                nfr = torch.clamp((nv + 319) // 320, min=8, max=self.dims.n_audio_ctx)
                n_rows, pi, pj, plen = eng.dtw_launch(dec, self.tokenizer.eot, mode=1 if dtw == "inrepo" else 0, n_frames=nfr)
                host["n_rows"][:n].copy_(n_rows[:n], non_blocking=True)
                host["pi"][:n].copy_(pi[:n], non_blocking=True)
                host["pj"][:n].copy_(pj[:n], non_blocking=True)
                host["plen"][:n].copy_(plen[:n], non_blocking=True)
            mark(4)
            slot.marks = marks
            slot.event.record(st)

    def _finish_pass(self, slot, language, dtw):
        """host half of a pass: waits for the slot's copies, then token lists -> text (+ DTW words)"""
        slot.event.synchronize()
        if getattr(slot, "marks", None) and self.stage_ms is not None:
            for k, name in enumerate(("logmel", "encode", "decode", "dtw")):
                self.stage_ms[name] = self.stage_ms.get(name, 0.0) + slot.marks[k].elapsed_time(slot.marks[k + 1])
        tok = self.tokenizer
        h = slot.host(slot.n)
        toks, slp, nsp = h["tokens"].numpy(), h["sum_lp"].numpy(), h["nsp"].numpy()
        out = []
        for i in range(slot.n):
            seq = toks[i, slot.n_prompt: slot.n_prompt + slot.n_sampled].tolist()
            if tok.eot in seq:
                seq = seq[: seq.index(tok.eot)]
            text_ids = [t for t in seq if t < tok.eot]
            text = tok.decode(text_ids).strip()
            r = {"tokens": seq, "text": text, "avg_logprob": float(slp[i]) / (len(seq) + 1),
                 "sum_logprob": float(slp[i]), "no_speech_prob": float(nsp[i]), "language": language,
                 "compression_ratio": _compression_ratio(text)}
            if dtw:
                L = int(h["plen"][i])
                path = np.stack([h["pi"][i, :L].numpy()[::-1], h["pj"][i, :L].numpy()[::-1]]).astype(np.int32)
                info = (int(h["n_rows"][i]), path)
                sp = []
                words = self._dtw_words_inrepo(text_ids, info) if dtw == "inrepo" else self._dtw_words(text_ids, info, sp)
                # per kept word (whitespace-only words are dropped): the index just behind its last token in the chunk's
                # TEXT ids (tokens < eot, timestamps excluded) -- what the multi-GPU record carries beside the times
                r["word_tok_end"] = [w.pop("tok_end") for w in words]
                r["words"] = words
                if sp:
                    r["word_spans_np"] = sp[0]       # the same three columns as int32 arrays (parallel.pack_records takes them as they are)
            out.append(r)
        return out

    def _decode_chunks(self, chunks: List[Any], language: Optional[str], task: str, word_timestamps, **kw):
        """see _decode_chunks_locked; calls from several user threads take turns (the engine contexts belong to one
        scheduler run at a time)"""
        # The host halves of the passes allocate a few hundred thousand small objects (token lists, word dicts) and no
        # reference cycles; left on, the cyclic collector runs full collections of the whole heap in the middle of the job,
        # on whichever launcher thread trips the threshold (the alignment stage measured 50 ms against 100 with and without
        # them, tools/prof_align_stage.py).  Paused for the duration of a scheduler run, restored whatever happens.
        import gc
        was_on = gc.isenabled()
        gc.disable()
        try:
            with self._call_lock:
                return self._decode_chunks_locked(chunks, language, task, word_timestamps, **kw)
        finally:
            if was_on:
                gc.enable()

    def _decode_chunks_locked(self, chunks: List[Any], language: Optional[str], task: str, word_timestamps,
                              forced_len: int = 0, passes_in_flight: Optional[int] = None, rows_per_pass: Optional[int] = None,
                              _force_split: int = 0, forced_lens=None, pass_rows: Optional[List[int]] = None):
        """chunks: list of <= 30 s float32 arrays (numpy, or torch tensors already resident in HBM) -> list of dicts
        {tokens, text, avg_logprob, ...} in input order.

        The scheduler of the hot path: the chunk list is cut into passes of `rows_per_pass` rows; up to
        `passes_in_flight` engine contexts, each with its own HIP stream and host launcher thread, carry consecutive
        passes concurrently (a decode step is a chain of ~300 short dependent kernels: independent passes fill each
        other's dependency gaps); a pass's results land in pinned host buffers at its end and are turned into text by
        its launcher thread while the thread's next pass runs.  Rows are independent and the kernels keep a fixed
        summation order, so the tokens do not depend on the number of passes in flight
        (tests/test_gpu_backend.py::test_scheduler_tokens_equal_single_engine)."""
        tok = self.tokenizer
        if language is None:
            language = self.detect_language(chunks[0]) if self.is_multilingual else "en"
        prompt = tok.sot_sequence(language, task)
        dtw = {True: "upstream", "dtw": "upstream", "dtw_inrepo": "inrepo"}.get(word_timestamps, False)
        if dtw and self.dtw_variant == "inrepo":
            dtw = "inrepo"
        # longest chunks first (stable: fixed 30 s windows keep their order): a pass decodes until its longest row has
        # ended, and token count follows speech duration (r = 0.79 over the reference's 81 VAD windows), so passes of
        # similar durations waste fewer steps -- 815 instead of 964 decode steps for those windows in passes of 16.
        # Rows are independent: the order changes no token.  Results are handed back in input order.
        order = sorted(range(len(chunks)), key=lambda i: -int(chunks[i].shape[0] if hasattr(chunks[i], "shape") else len(chunks[i])))
        in_order = order == list(range(len(chunks)))
        chunks_in = chunks                      # the fall-back calls below start over from the caller's order
        if not in_order:
            chunks = [chunks[i] for i in order]
        flens = None if forced_lens is None else [forced_lens[i] for i in order]
        # the cut (scheduler.plan_job): the default scheduler merges requests into wide passes; rows_per_pass /
        # passes_in_flight / pass_rows pin it.  `plan.R` is the launch shape of THIS cut (after any re-plan for fewer lanes)
        plan = plan_job(len(chunks), self.engine.max_batch, lambda R, need: self._default_lanes(R, need=need),
                        auto_rows=self.auto_rows, rows_per_pass=rows_per_pass, passes_in_flight=passes_in_flight,
                        pass_rows=pass_rows, default_rows=self.rows_per_pass,
                        launch_bound=self.dims.n_text_state <= 512)        # tiny / base: a decode step is a launch chain (scheduler.plan_passes)
        sizes, lanes = plan.sizes, plan.lanes
        n_pass = len(sizes)
        self.last_plan = plan.report()
        passes, a = [], 0
        for sz in sizes:
            passes.append(chunks[a: a + sz])
            a += sz
        pass_start = [sum(sizes[:i]) for i in range(n_pass)]
        engines = self._get_engines(max(1, min(lanes, len(passes))), rows=plan.R)
        n_eng = len(engines)
        self.last_plan["passes_in_flight"] = max(1, min(n_eng, n_pass))      # (fewer when a context did not fit after all)
        # one key split for every pass size: the split fixes the summation order of the cross-attention, so tokens do
        # not depend on how the scheduler cuts the chunk list (48-row passes would be 0.7 % faster without a split)
        cross_split = _force_split or self.cross_split or 2
        # several passes in flight: the K = 4d GEMV as 80 fat blocks (leaves CUs to the other passes)
        fc2_tile_n = self.fc2_tile_n if self.fc2_tile_n is not None else (16 if n_eng > 1 else 0)
        # One launch shape per context: a pass of fewer rows is launched with R rows, the extra ones being padding that
        # counts as finished (wx_decode_opts.n_active) -- hipGraphs are captured per row count and a new count costs
        # ~80 ms per context (tools/cold_shape_cost.py: 320 ms on the first job with a new remainder).  Passes of more than
        # 16 rows launch whole 16-row groups (the GEMV kernels walk those): at most R / 16 shapes.
        launch_rows = plan.launch_rows

        results: List[Any] = [None] * len(passes)
        errors: List[BaseException] = []
        job = PassJob(passes=passes, pass_start=pass_start, prompt=prompt, dtw=dtw, forced_len=forced_len, cross_split=cross_split,
                      fc2_tile_n=fc2_tile_n, flens=flens, launch_rows=launch_rows, language=language, results=results, errors=errors)
        lanes = [Lane(self, e, job) for e in engines]
        todo = [list(range(k, len(passes), n_eng)) for k in range(n_eng)]
        if n_eng == 1:
            lanes[0].run(todo[0])
        else:
            # hipGraph captures must not race with other threads' launches: the first pass of every launch shape an
            # engine has not captured yet (a full pass, a ragged last pass) is enqueued from this thread, engine
            # after engine, before the launcher threads start (an enqueue does not wait for the GPU to finish)
            # everything a captured decode step bakes in (api.hip: the graph key of wx_decode_greedy), as far as this
            # scheduler varies it: the step variant actually launched, and -- with per-row forced lengths -- which of the
            # context's two pass slots the pass uses (the slot's length buffer is a kernel argument)
            variant = 1 if cross_split != 2 else self.step_variant
            sig = (tuple(prompt), self.rules, forced_len, dtw, cross_split, fc2_tile_n, variant, self.profile_launches)
            try:
                for k, lane in enumerate(lanes):
                    for i in list(todo[k]):
                        key = sig + (launch_rows(len(passes[i])), (lane.j & 1) if flens is not None else -1)
                        if key not in lane.eng.warm and len(lane.pending) < 2:
                            lane.enqueue(i)
                            todo[k].remove(i)
                            lane.eng.warm.add(key)
            except BaseException as e:        # noqa: BLE001
                errors.append(e)
            if not errors:
                th = [threading.Thread(target=lanes[k].run, args=(todo[k], 32)) for k in range(n_eng)]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
        if errors:
            for eng in engines:
                torch.cuda.synchronize(eng.device)
            gave_up = [e for e in errors if isinstance(e, WxError) and "gave up" in str(e)]
            if gave_up and len(gave_up) == len(errors) and not _force_split:
                # The bounded wait of a key-split merge expired (one kernel per stage, step_variant 1: the last split of a
                # (row, head) waits for the other split's partials; the default fused launch has both splits in one block
                # and never raises the flag).  The rows were NaN-poisoned and the device flag raised, now cleared by
                # wx_device_status.  Decode this call again without key splits, where no block waits for another; stay
                # there only after it has happened three times.  Launch shapes captured so far do not cover the new
                # variant: forget them, so that the first pass of each is enqueued from this thread again.
                self.split_giveups += 1
                for eng in engines:
                    eng.warm.clear()
                warnings.warn("a decode kernel gave up waiting for a key split; decoding the batch again without key splits")
                if self.split_giveups >= 3:
                    self.cross_split = 1
                return self._decode_chunks_locked(chunks_in, language, task, word_timestamps, forced_len=forced_len,
                                                  passes_in_flight=passes_in_flight, rows_per_pass=rows_per_pass, _force_split=1,
                                                  forced_lens=forced_lens, pass_rows=pass_rows)
            raise errors[0]
        self.selfq_blocks += sum(l.selfq for l in lanes)
        flat = [r for p in results for r in p]
        if in_order:
            return flat
        out = [None] * len(flat)
        for k, i in enumerate(order):
            out[i] = flat[k]
        return out

    def _dtw_words(self, text_ids, path_info, spans=None):
        """published find_alignment bookkeeping (dtw_words.words_upstream)"""
        return DW.words_upstream(self.tokenizer, text_ids, path_info, spans)

    def _dtw_words_inrepo(self, text_ids, path_info):
        """the reference's own in-repo bookkeeping, mlx_whisper_optimized_final.py:215-251 (dtw_words.words_inrepo)"""
        return DW.words_inrepo(self.tokenizer, text_ids, path_info)

    def transcribe_batch(self, segments: List[Dict[str, Any]], batch_size: int = 8, align_words: bool = False,
                         language: Optional[str] = None, task: str = "transcribe", word_timestamps=False, **kwargs):
        """mlx_lightning.py:82-119 semantics on batches: each VAD segment (<= 30 s, longer ones are
        windowed) becomes result segments with absolute times clamped to the VAD segment."""
        chunks, owner = [], []
        for si, seg in enumerate(segments):
            audio = seg.get("audio")
            if audio is None:
                continue
            for off in range(0, max(len(audio), 1), N_SAMPLES):
                chunks.append(audio[off: off + N_SAMPLES])
                owner.append((si, off / SAMPLE_RATE, min(len(audio) - off, N_SAMPLES) / SAMPLE_RATE))
        dtw = word_timestamps if word_timestamps in ("dtw", "dtw_inrepo") else (word_timestamps is True and not align_words)
        # batch_size = chunks per request, as in the reference's call (asr.py:80-87).  With the default coalesce=None the
        # scheduler merges requests into passes of up to 128 rows when the job is large enough (plan_passes: rows are
        # independent, tokens do not change); load_model(..., coalesce=k) pins passes to batch_size * k rows instead
        fls = kwargs.get("forced_lens")          # bench workload: one length per segment (segments are not windowed further there)
        if fls is not None:
            assert len(fls) == len(chunks), "forced_lens: one entry per <= 30 s segment"
        results = self._decode_chunks(chunks, language, task or "transcribe", dtw, forced_len=int(kwargs.get("forced_len", 0)), forced_lens=fls,
                                      passes_in_flight=kwargs.get("passes_in_flight"), pass_rows=kwargs.get("pass_rows"),
                                      rows_per_pass=kwargs.get("rows_per_pass") or
                                      (None if self.auto_rows else (batch_size or self.max_batch) * self.coalesce)) if chunks else []
        all_segments = []
        lang = None
        for (si, off, dur), r in zip(owner, results):
            lang = lang or r["language"]
            if not r["text"]:
                continue
            seg = segments[si]
            s = {"start": seg["start"] + off, "end": min(seg["start"] + off + dur, seg["end"]), "text": r["text"],
                 "id": len(all_segments)}
            if dtw:
                s["words"] = [dict(w, start=w["start"] + s["start"], end=min(w["end"] + s["start"], s["end"]))
                              for w in r.get("words", [])]
            all_segments.append(s)
        result = {"segments": all_segments, "language": lang or language or "en"}
        if kwargs.get("return_chunks"):
            # per-chunk records (tokens, log-probabilities, word spans) in input order: what the multi-GPU gather packs
            result["chunks"] = [dict(r, segment=si, offset=off) for (si, off, _d), r in zip(owner, results)]
        if align_words and segments:
            result = self._align_batch_words(result, segments)
        return result

    def transcribe(self, audio: Union[str, np.ndarray], batch_size: Optional[int] = None, num_workers: int = 0,
                   language: Optional[str] = None, task: Optional[str] = None, chunk_size: int = 30,
                   print_progress: bool = False, combined_progress: bool = False, verbose: bool = False,
                   align_words: bool = False, **kwargs):
        """mlx_lightning.py:121-227: consecutive 30 s windows, one result segment per window."""
        kwargs.pop("batch_size", None)
        if isinstance(audio, str):
            audio = load_audio(audio)
        audio = np.asarray(audio, dtype=np.float32)
        segs = [{"start": s / SAMPLE_RATE, "end": min(s + N_SAMPLES, len(audio)) / SAMPLE_RATE,
                 "audio": audio[s: s + N_SAMPLES]} for s in range(0, max(len(audio), 1), N_SAMPLES)]
        res = self.transcribe_batch(segs, batch_size=batch_size or self.max_batch, language=language,
                                    task=task or "transcribe", align_words=False, **kwargs)
        res["text"] = " ".join(s["text"] for s in res["segments"])
        if align_words:
            res = self._align_words(res, audio, language)
        return res

    # ------------------------------------------------------------------ word alignment (wav2vec2)
    def _get_align_model(self, language):
        from . import alignment
        key = f"align_{language}"
        if key not in self.align_model_cache:
            self.align_model_cache[key] = alignment.load_align_model(language, f"cuda:{self.device_index}",
                                                                     model_dir=self.align_model_dir)
        return self.align_model_cache[key]

    def _align_words(self, transcription_result, audio, language=None):
        """mlx_lightning.py:229-288 (degrades to a warning, :279-286)."""
        from . import alignment
        try:
            language = language or transcription_result.get("language", "en")
            model_a, metadata = self._get_align_model(language)
            aligned = alignment.align(transcription_result["segments"], model_a, metadata, audio,
                                      f"cuda:{self.device_index}", return_char_alignments=False)
            transcription_result["segments"] = aligned["segments"]
            for segment in transcription_result["segments"]:
                segment.setdefault("words", [])
                for word in segment["words"]:
                    if "score" in word and "probability" not in word:
                        word["probability"] = word["score"]
        except Exception as e:
            print(f"Warning: Word alignment failed: {e}")
            print("Returning transcription without word-level timestamps")
            for segment in transcription_result["segments"]:
                segment.setdefault("words", [])
        return transcription_result

    @staticmethod
    def _group_by_vad(result_segments, segments):
        """mlx_lightning.py:300-317: the result segments that lie inside each VAD segment, relative to its start --
        [(index of the VAD segment, [relative segment dicts])] for the VAD segments that got any"""
        groups, idx = [], 0
        for vi, vad in enumerate(segments):
            if vad.get("audio") is None:
                continue
            rel = []
            while idx < len(result_segments):
                seg = result_segments[idx]
                if seg["start"] >= vad["start"] and seg["end"] <= vad["end"]:
                    c = dict(seg)
                    c["start"] -= vad["start"]
                    c["end"] -= vad["start"]
                    rel.append(c)
                    idx += 1
                else:
                    break
            if rel:
                groups.append((vi, rel))
        return groups

    @staticmethod
    def _offset_aligned(aligned, vad_start):
        """mlx_lightning.py:330-345: an align() result of one VAD segment back on the file's time axis (in place)"""
        out = []
        for a in aligned.get("segments", []):
            a["start"] += vad_start
            a["end"] += vad_start
            for w in a.get("words", []):
                if "start" in w:
                    w["start"] += vad_start
                if "end" in w:
                    w["end"] += vad_start
            out.append(a)
        return out

    def align_groups(self, groups, segments, language, _trace=None):
        """every VAD segment's transcript aligned against its own audio, all of them in ONE batched call (the reference
        loops align() over the VAD segments, mlx_lightning.py:318-328): list of align() result dicts, relative times"""
        from . import alignment
        model_a, metadata = self._get_align_model(language)
        return alignment.align_batch([(rel, segments[vi]["audio"]) for vi, rel in groups], model_a, metadata,
                                     f"cuda:{self.device_index}", _trace=_trace)

    def _align_batch_words(self, result, segments):
        """mlx_lightning.py:290-369, with the transcripts of ALL VAD segments aligned in one batched call."""
        try:
            language = result.get("language", "en")
            groups = self._group_by_vad(result["segments"], segments)
            aligned_segments = []
            for (vi, _rel), aligned in zip(groups, self.align_groups(groups, segments, language)):
                aligned_segments += self._offset_aligned(aligned, segments[vi]["start"])
            result["segments"] = aligned_segments
        except Exception as e:
            print(f"Warning: Batch word alignment failed: {e}")
            print("Returning transcription without word-level timestamps")
        return result

    # ------------------------------------------------------------------ misc
    def detect_language(self, audio: np.ndarray) -> str:
        """mlx_lightning.py:371-390: first 30 s, argmax over the language tokens after <|sot|>."""
        if not self.is_multilingual:
            return "en"
        with self._call_lock:
            return self._detect_language_locked(audio)

    def _detect_language_locked(self, audio) -> str:
        eng, tok = self.engine, self.tokenizer
        if torch.is_tensor(audio):             # chunks may already be resident in HBM (transcribe_batch on device tensors)
            c = audio.reshape(-1)[:N_SAMPLES].to(device=eng.device, dtype=torch.float32)
        else:
            c = torch.from_numpy(np.asarray(audio, dtype=np.float32).reshape(-1)[:N_SAMPLES])
        pcm = torch.zeros(1, N_SAMPLES, dtype=torch.float32, device=eng.device)
        pcm[0, : len(c)] = c
        enc = eng.encode(eng.logmel(pcm, torch.tensor([len(c)], dtype=torch.int32)))
        logits = eng.decode_logits(enc, torch.tensor([[tok.sot]], dtype=torch.int32))
        lang = logits[0, tok.sot + 1: tok.sot + 1 + tok.n_langs].cpu().numpy()
        return LANGUAGES[int(np.argmax(lang))]

    @property
    def supported_languages(self) -> List[str]:
        return LANGUAGES[: self.tokenizer.n_langs] if self.is_multilingual else ["en"]

    @property
    def is_multilingual(self) -> bool:
        return not self.model_name.endswith(".en")


def _compression_ratio(text: str) -> float:
    """mlx_whisper_batch_decoder.py:470-477"""
    b = text.encode("utf-8")
    try:
        return len(b) / len(zlib.compress(b))
    except Exception:
        return 1.0


class HipWhisperPipeline:
    """Stand-in for MLXWhisperPipeline (whisperx/asr.py:19-147): optional VAD callable that returns
    merged chunks [{"start","end","segments"}], then backend.transcribe_batch."""

    def __init__(self, backend, vad_model=None):
        self.backend = backend
        self.vad_model = vad_model

    def transcribe(self, audio, batch_size: int = 8, chunk_size: int = 30, print_progress=False,
                   combined_progress=False, verbose=False, **kwargs):
        if isinstance(audio, str):
            audio = load_audio(audio)
        if hasattr(self.backend, "_align_words") and kwargs.get("word_timestamps", False) is True:
            kwargs["align_words"] = True          # asr.py:50-52,76-78
            kwargs.pop("word_timestamps", None)
        if self.vad_model is None:
            return self.backend.transcribe(audio, batch_size=batch_size, num_workers=0, print_progress=print_progress,
                                           combined_progress=combined_progress, verbose=verbose, **kwargs)
        segments = self.vad_model(audio, chunk_size)
        for seg in segments:
            seg["audio"] = audio[int(seg["start"] * SAMPLE_RATE): int(seg["end"] * SAMPLE_RATE)]   # asr.py:70-73
        return self.backend.transcribe_batch(segments, batch_size=batch_size, print_progress=print_progress,
                                             combined_progress=combined_progress, verbose=verbose, **kwargs)

    def detect_language(self, audio):
        return self.backend.detect_language(audio)


from .vad import SileroVad, merge_chunks      # noqa: E402,F401  (re-exported: backend.merge_chunks is the round-1 name)


def load_model(whisper_arch: str, device: str = "cuda", device_index: int = 0, compute_type: str = "float16",
               asr_options: Optional[dict] = None, language: Optional[str] = None, vad_method: Optional[str] = None,
               vad_options: Optional[dict] = None, task: str = "transcribe", download_root: Optional[str] = None,
               local_files_only: bool = False, threads: int = 4, backend: str = "hip", batch_size: int = 16,
               vad_model=None, **kwargs):
    """whisperx/asr.py:150-275 for backend in ("hip", "mi355x").  VAD front-ends are CPU pre-steps outside this
    path: `vad_method="silero"` builds the reference's Silero front-end (whisperx_mlx_amd/vad.py; offline: from the
    torch.hub cache or vad_options["repo_dir"]), `vad_model=callable(audio, chunk_size)` takes any producer of merged
    chunks, and with neither the audio is cut into fixed 30 s windows (pyannote is out of scope)."""
    if backend not in ("hip", "mi355x", "auto"):
        raise ValueError(f"this package only provides backend='hip' (got {backend!r})")
    if vad_model is None and vad_method == "silero":
        vo = {"vad_onset": 0.500, "vad_offset": 0.363, **(vad_options or {})}        # asr.py:241-245
        vad_model = SileroVad.from_hub(vo.get("repo_dir"), vo["vad_onset"], vo["vad_offset"])
    elif vad_model is None and vad_method not in (None, "none"):
        raise ValueError(f"vad_method {vad_method!r} is not provided by this package (use 'silero', 'none' or vad_model=)")
    kwargs.pop("word_timestamps", None)           # asr.py:200
    be = WhisperHipBackend(whisper_arch, device=device, device_index=device_index, compute_type=compute_type,
                           download_root=download_root, local_files_only=local_files_only, threads=threads,
                           max_batch=min(max(batch_size, 1), 64), **kwargs)     # kwargs: coalesce, passes_in_flight, dtw_variant
    return HipWhisperPipeline(be, vad_model)
