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
import os
import subprocess
import zlib
from abc import ABC, abstractmethod
from typing import Any, Dict, List, Optional, Union

import numpy as np
import torch

from . import weights as W
from .audio import N_SAMPLES, SAMPLE_RATE, TOKENS_PER_SECOND
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


class WhisperHipBackend(WhisperBackend):
    def __init__(self, model: str, device: str = "cuda", device_index: int = 0, compute_type: str = "float16",
                 download_root: Optional[str] = None, local_files_only: bool = False, threads: int = 4,
                 max_batch: int = 16, random_init: bool = False, seed: int = 0, **kwargs):
        if compute_type not in ("float16", "fp16", "default", "int8"):
            raise ValueError(f"backend 'hip' computes in float16, optionally with int8 decoder weights (got compute_type={compute_type!r})")
        self.model_name = model
        # "int8": the decode GEMV weights are stored as int8 + row scales and dequantised in registers; activations,
        # accumulation, the encoder and the cross K/V projection stay as in float16 (weights.quantize_packed_decoder)
        self.compute_type = "int8" if compute_type == "int8" else "float16"
        self.device_index = device_index
        name = W.resolve_model_name(model)
        key = f"{model}|{device_index}|{download_root}|{random_init}|{seed}|{max_batch}|{self.compute_type}"
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
                sd = W.random_checkpoint(dims, seed=seed, device=dev)
            else:
                raise FileNotFoundError(
                    f"no local checkpoint for {model!r} (looked in {download_root!r}); the GPU box has no network. "
                    "Pass download_root=<dir with config.json + safetensors> or random_init=True for a throughput run")
            heads = extra.get("alignment_heads") or W.default_alignment_heads(name, dims)
            packed = W.pack(sd, dims, dev)
            if self.compute_type == "int8":
                packed = W.quantize_packed_decoder(packed, dims)
            eng = WhisperHipEngine(dims, packed, max_batch=max_batch, device_index=device_index, alignment_heads=heads)
            _engine_cache[key] = (eng, dims, ckpt_dir, extra)
        self.engine, self.dims, self.ckpt_dir, self.extra = _engine_cache[key]
        self.max_batch = max_batch
        self.temperature = kwargs.get("temperature", 0.0)      # greedy only (mlx_lightning.py:77)
        self.tokenizer = get_tokenizer(self.dims.n_vocab, model_dir=self.ckpt_dir)
        self.suppress = self.tokenizer.suppress_tokens(self.extra.get("suppress_tokens"))
        self.rules = kwargs.get("rules", RULES_LIGHTNING)
        self.align_model_cache = {}
        self.align_model_dir = kwargs.get("align_model_dir", download_root)

    # ------------------------------------------------------------------ core
    def _decode_chunks(self, chunks: List[np.ndarray], language: Optional[str], task: str, word_timestamps: bool,
                       forced_len: int = 0):
        """chunks: list of <= 30 s float32 arrays -> list of dicts {tokens, text, avg_logprob, ...}"""
        eng, tok = self.engine, self.tokenizer
        if language is None:
            language = self.detect_language(chunks[0]) if self.is_multilingual else "en"
        prompt = tok.sot_sequence(language, task)
        out = []
        B = self.max_batch
        for b0 in range(0, len(chunks), B):
            batch = chunks[b0: b0 + B]
            n = len(batch)
            pcm = torch.zeros(n, N_SAMPLES, dtype=torch.float32)
            nv = torch.zeros(n, dtype=torch.int32)
            for i, c in enumerate(batch):
                c = np.asarray(c, dtype=np.float32)[:N_SAMPLES]
                pcm[i, : len(c)] = torch.from_numpy(c)
                nv[i] = len(c)
            mel = eng.logmel(pcm.to(eng.device), nv.to(eng.device))
            enc = eng.encode(mel)
            dec = eng.decode(enc, tok, prompt, rules=self.rules, suppress_ids=self.suppress,
                             capture_qk=word_timestamps, forced_len=forced_len)
            # published find_alignment crops the attention to the frames that carry audio (num_frames // 2)
            nfr = torch.clamp((nv + 319) // 320, min=8, max=self.dims.n_audio_ctx)
            paths = eng.dtw_path(dec, tok.eot, n_frames=nfr) if word_timestamps else None
            eng.check_status()    # raises if a kernel's bounded wait gave up (results would be poisoned)
            toks = dec.tokens.cpu().numpy()
            slp = dec.sum_logprob.cpu().numpy()
            nsp = dec.no_speech_prob.cpu().numpy()
            for i in range(n):
                seq = toks[i, dec.n_prompt: dec.n_prompt + dec.n_sampled].tolist()
                if tok.eot in seq:
                    seq = seq[: seq.index(tok.eot)]
                text_ids = [t for t in seq if t < tok.eot]
                text = tok.decode(text_ids).strip()
                r = {"tokens": seq, "text": text, "avg_logprob": float(slp[i]) / (len(seq) + 1),
                     "no_speech_prob": float(nsp[i]), "language": language,
                     "compression_ratio": _compression_ratio(text)}
                if word_timestamps:
                    r["words"] = self._dtw_words(text_ids, paths[i])
                out.append(r)
        return out

    def _dtw_words(self, text_ids, path_info):
        """word times from the DTW path over the alignment matrix rows (text tokens + EOT):
        published find_alignment bookkeeping (jumps of the token index -> first frame)."""
        n_rows, path = path_info
        if n_rows < 2 or path.shape[1] == 0 or not text_ids:
            return []
        ti, fi = path[0], path[1]
        jumps = np.concatenate([[True], np.diff(ti) > 0])
        jump_times = fi[jumps].astype(np.float64) / TOKENS_PER_SECOND
        words, word_tokens = self.tokenizer.split_to_word_tokens(text_ids)
        bounds = np.concatenate([[0], np.cumsum([len(t) for t in word_tokens])])
        n_text = min(len(text_ids), len(jump_times) - 1) if len(jump_times) > 1 else len(jump_times)
        res = []
        for w, a, b in zip(words, bounds[:-1], bounds[1:]):
            if a >= len(jump_times):
                break
            start = float(jump_times[a])
            end = float(jump_times[min(b, len(jump_times) - 1)])
            if w.strip():
                res.append({"word": w.strip(), "start": start, "end": max(end, start), "probability": 1.0})
        return res

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
        dtw = word_timestamps == "dtw" or (word_timestamps is True and not align_words)
        results = self._decode_chunks(chunks, language, task or "transcribe", dtw) if chunks else []
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

    def _align_batch_words(self, result, segments):
        """mlx_lightning.py:290-369, with every VAD segment's transcript aligned in one batched call."""
        from . import alignment
        try:
            language = result.get("language", "en")
            model_a, metadata = self._get_align_model(language)
            aligned_segments, idx = [], 0
            for vad in segments:
                if vad.get("audio") is None:
                    continue
                rel = []
                while idx < len(result["segments"]):
                    seg = result["segments"][idx]
                    if seg["start"] >= vad["start"] and seg["end"] <= vad["end"]:
                        c = dict(seg)
                        c["start"] -= vad["start"]
                        c["end"] -= vad["start"]
                        rel.append(c)
                        idx += 1
                    else:
                        break
                if rel:
                    aligned = alignment.align(rel, model_a, metadata, vad["audio"], f"cuda:{self.device_index}")
                    for a in aligned.get("segments", []):
                        a["start"] += vad["start"]
                        a["end"] += vad["start"]
                        for w in a.get("words", []):
                            if "start" in w:
                                w["start"] += vad["start"]
                            if "end" in w:
                                w["end"] += vad["start"]
                        aligned_segments.append(a)
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
        eng, tok = self.engine, self.tokenizer
        c = np.asarray(audio, dtype=np.float32)[:N_SAMPLES]
        pcm = torch.zeros(1, N_SAMPLES, dtype=torch.float32)
        pcm[0, : len(c)] = torch.from_numpy(c)
        enc = eng.encode(eng.logmel(pcm.to(eng.device), torch.tensor([len(c)], dtype=torch.int32)))
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


def merge_chunks(segments, chunk_size, onset: float = 0.5, offset: Optional[float] = 0.363):
    """whisperx/vads/vad.py:20-53: merge VAD speech turns [(start, end), ...] into chunks of at
    most `chunk_size` seconds."""
    curr_end = 0
    merged, seg_idxs = [], []
    curr_start = segments[0][0]
    for (s, e) in segments:
        if e - curr_start > chunk_size and curr_end - curr_start > 0:
            merged.append({"start": curr_start, "end": curr_end, "segments": seg_idxs})
            curr_start = s
            seg_idxs = []
        curr_end = e
        seg_idxs.append((s, e))
    merged.append({"start": curr_start, "end": curr_end, "segments": seg_idxs})
    return merged


def load_model(whisper_arch: str, device: str = "cuda", device_index: int = 0, compute_type: str = "float16",
               asr_options: Optional[dict] = None, language: Optional[str] = None, vad_method: Optional[str] = None,
               vad_options: Optional[dict] = None, task: str = "transcribe", download_root: Optional[str] = None,
               local_files_only: bool = False, threads: int = 4, backend: str = "hip", batch_size: int = 16,
               vad_model=None, **kwargs):
    """whisperx/asr.py:150-275 for backend in ("hip", "mi355x").  VAD front-ends (silero /
    pyannote) are CPU pre-steps outside this path: pass `vad_model=callable(audio, chunk_size)`
    returning merged chunks (see merge_chunks) or leave it None for fixed 30 s windows."""
    if backend not in ("hip", "mi355x", "auto"):
        raise ValueError(f"this package only provides backend='hip' (got {backend!r})")
    kwargs.pop("word_timestamps", None)           # asr.py:200
    be = WhisperHipBackend(whisper_arch, device=device, device_index=device_index, compute_type=compute_type,
                           download_root=download_root, local_files_only=local_files_only, threads=threads,
                           max_batch=min(max(batch_size, 1), 64), **kwargs)
    return HipWhisperPipeline(be, vad_model)
