"""CLI task: flags -> load_model -> transcribe -> align -> writers.

Mirrors the observable flow of the reference's `whisperx/transcribe.py:17-250` (`transcribe_task`) and the
flag names of `whisperx/__main__.py:17-88` for the pieces this package provides; Whisper and wav2vec2 run on
the HIP backend.  Out of scope here, as in SURVEY 8: speaker diarization (`--diarize` is rejected), the
neural VAD front-ends (`--vad_method none` = fixed 30 s windows; "silero" is used only if `torch.hub` has it
cached locally), sampling / beam search (`--temperature`, `--beam_size`: the reference's MLX backends are
greedy too, `mlx_whisper_batch_decoder.py:267-303`).
"""
import argparse
import gc
import os
import warnings
from typing import Optional

import numpy as np

from .tokenizer import LANGUAGES, TO_LANGUAGE_CODE
from .writers import get_writer


def str2bool(s):
    if s in ("True", "False"):
        return s == "True"
    raise ValueError(f"Expected one of {{'True', 'False'}}, got {s}")


def optional_int(s):
    return None if s == "None" else int(s)


def optional_float(s):
    return None if s == "None" else float(s)


def build_parser() -> argparse.ArgumentParser:
    # fmt: off
    p = argparse.ArgumentParser(prog="whisperx_mlx_amd", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("audio", nargs="+", type=str, help="audio file(s) to transcribe")
    p.add_argument("--model", default="small", help="Whisper model name (tiny ... large-v3, large-v3-turbo, distil-large-v3)")
    p.add_argument("--backend", default="hip", choices=["auto", "hip", "mi355x"], help="transcription backend (this package: HIP on MI355X)")
    p.add_argument("--model_cache_only", type=str2bool, default=True, help="never download: checkpoints are read from --model_dir")
    p.add_argument("--model_dir", type=str, default=None, help="directory holding the Whisper (and wav2vec2) checkpoints")
    p.add_argument("--random_init", type=str2bool, default=False, help="seeded random weights of the named architecture (throughput runs)")
    p.add_argument("--device", default="cuda", help="device (the HIP backend needs a GPU)")
    p.add_argument("--device_index", default=0, type=int)
    p.add_argument("--batch_size", default=16, type=int, help="chunks per pass of the hot path (<= 48)")
    p.add_argument("--compute_type", default="float16", type=str, choices=["float16", "int8"], help="float16: fp16 storage, fp32 accumulation; int8: int8 decoder GEMV weights (+ row scales), everything else as float16")
    p.add_argument("--word_timestamps", type=str2bool, default=False, help="word times from the decoder's cross-attention (DTW) instead of wav2vec2 alignment")
    p.add_argument("--output_dir", "-o", type=str, default=".")
    p.add_argument("--output_format", "-f", type=str, default="all", choices=["all", "srt", "vtt", "txt", "tsv", "json", "aud"])
    p.add_argument("--verbose", type=str2bool, default=True)
    p.add_argument("--task", type=str, default="transcribe", choices=["transcribe", "translate"])
    p.add_argument("--language", type=str, default=None, help="language spoken in the audio; None = detect")
    # alignment
    p.add_argument("--align_model", default=None, help="wav2vec2 CTC checkpoint (name under --model_dir or a directory)")
    p.add_argument("--interpolate_method", default="nearest", choices=["nearest", "linear", "ignore"])
    p.add_argument("--no_align", action="store_true", help="do not run the wav2vec2 forced alignment")
    p.add_argument("--return_char_alignments", action="store_true")
    # VAD
    p.add_argument("--vad_method", type=str, default="none", choices=["none", "silero"], help="'none': fixed windows of --chunk_size seconds")
    p.add_argument("--vad_onset", type=float, default=0.500)
    p.add_argument("--vad_offset", type=float, default=0.363)
    p.add_argument("--chunk_size", type=int, default=30)
    # accepted for command-line compatibility
    p.add_argument("--diarize", action="store_true", help="not provided by this package")
    p.add_argument("--temperature", type=float, default=0)
    p.add_argument("--beam_size", type=optional_int, default=1)
    p.add_argument("--suppress_tokens", type=str, default="-1")
    p.add_argument("--initial_prompt", type=str, default=None)
    p.add_argument("--max_line_width", type=optional_int, default=None)
    p.add_argument("--max_line_count", type=optional_int, default=None)
    p.add_argument("--highlight_words", type=str2bool, default=False)
    p.add_argument("--threads", type=optional_int, default=0)
    p.add_argument("--print_progress", type=str2bool, default=False)
    # fmt: on
    return p


def _normalise_language(lang: Optional[str]) -> Optional[str]:
    if lang is None:
        return None
    lang = lang.lower()
    if lang in LANGUAGES:
        return lang
    if lang in TO_LANGUAGE_CODE:
        return TO_LANGUAGE_CODE[lang]
    raise ValueError(f"Unsupported language: {lang}")


def transcribe_task(args: dict, parser: argparse.ArgumentParser):
    from .alignment import align, load_align_model
    from .backend import load_audio, load_model

    if args.pop("diarize"):
        parser.error("--diarize: speaker diarization is outside this package (SURVEY 8: out of scope)")
    if args["temperature"] not in (0, 0.0) or (args["beam_size"] or 1) > 1:
        warnings.warn("the HIP backend decodes greedily (temperature 0, no beam search); --temperature / --beam_size are ignored")
    output_dir = args.pop("output_dir")
    os.makedirs(output_dir, exist_ok=True)
    language = _normalise_language(args["language"])
    model_name = args["model"]
    if model_name.endswith(".en") and language != "en":
        if language is not None:
            warnings.warn(f"{model_name} is an English-only model but received '{language}'; using English instead.")
        language = "en"
    align_language = language if language is not None else "en"
    no_align = args["no_align"] or args["task"] == "translate" or args["word_timestamps"]   # translation cannot be aligned
    word_options = ("highlight_words", "max_line_count", "max_line_width")
    if no_align and not args["word_timestamps"]:
        for o in word_options:
            if args[o]:
                parser.error(f"--{o} not possible with --no_align")
    if args["max_line_count"] and not args["max_line_width"]:
        warnings.warn("--max_line_count has no effect without --max_line_width")
    writer = get_writer(args["output_format"], output_dir)
    writer_args = {o: args[o] for o in word_options}
    if (args["threads"] or 0) > 0:
        import torch
        torch.set_num_threads(args["threads"])

    vad = None
    if args["vad_method"] == "silero":
        from .vad import SileroVad
        vad = SileroVad.from_hub(None, args["vad_onset"], args["vad_offset"])
    device = args["device"] if ":" in args["device"] else f"{args['device']}:{args['device_index']}"
    model = load_model(model_name, device=args["device"], device_index=args["device_index"], compute_type=args["compute_type"],
                       language=language, task=args["task"], download_root=args["model_dir"],
                       local_files_only=args["model_cache_only"], backend=args["backend"], batch_size=args["batch_size"],
                       vad_model=vad, random_init=args["random_init"])
    results = []
    audio = None
    for audio_path in args["audio"]:
        audio = load_audio(audio_path)
        if args["verbose"]:
            print(">>Performing transcription...")
        result = model.transcribe(audio, batch_size=args["batch_size"], chunk_size=args["chunk_size"],
                                  print_progress=args["print_progress"], verbose=args["verbose"], language=language,
                                  task=args["task"], word_timestamps="dtw" if args["word_timestamps"] else False)
        results.append((result, audio_path))
    del model
    gc.collect()

    if not no_align:
        pending, results = results, []
        align_model, meta = load_align_model(align_language, device, model_name=args["align_model"], model_dir=args["model_dir"])
        for result, audio_path in pending:
            input_audio = audio_path if len(pending) > 1 else audio
            if len(result["segments"]) > 0:
                if result.get("language", "en") != meta["language"]:
                    print(f"New language found ({result['language']})! Previous was ({meta['language']}), "
                          "loading new alignment model for new language...")
                    align_model, meta = load_align_model(result["language"], device, model_dir=args["model_dir"])
                if args["verbose"]:
                    print(">>Performing alignment...")
                lang = result.get("language")
                result = align(result["segments"], align_model, meta, input_audio, device,
                               interpolate_method=args["interpolate_method"],
                               return_char_alignments=args["return_char_alignments"], print_progress=args["print_progress"])
                if lang is not None:
                    result["language"] = lang
            results.append((result, audio_path))
        del align_model
        gc.collect()

    written = []
    for result, audio_path in results:
        result = dict(result)
        if "language" not in result:
            result["language"] = align_language
        writer(result, audio_path, writer_args)
        written.append(audio_path)
    return written


def cli(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv).__dict__
    return transcribe_task(args, parser)
