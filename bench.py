#!/usr/bin/env python3
"""bench.py -- real-time factor of the HIP hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the whole hot path over one batch of B=16 synthetic 30 s chunks
already resident in HBM: log-mel -> Whisper encoder -> cross-KV projection -> greedy
decode (device-resident loop, forced token count) -> cross-attention DTW.  Weights are
seeded N(0, 0.02^2) fp16 in the exact large-v3 shapes (no checkpoint ships with the
reference), so the decode length is forced to the reference's measured mean
(145 sampled + 3 prompt tokens).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Chunks shard over ranks with no data-path collective; one RCCL all_gather of the fixed
size result records closes the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = 2500.0      # dense fp16/bf16 MFMA


def algorithmic_bytes(dims, B, kind, t_self=75):
    """fp16 bytes one launch must move."""
    d = dims.n_text_state
    if kind == "cross_attn":        # one layer: K and V of every sequence, read once
        return B * 2 * dims.n_audio_ctx * d * 2
    if kind == "decode_step":       # weights once + cross KV + self KV at t_self
        L = dims.n_text_layer
        w = 2 * (L * 14 * d * d + dims.n_vocab * d)
        return w + B * L * 2 * dims.n_audio_ctx * d * 2 + B * L * 2 * t_self * d * 2
    raise ValueError(kind)


def encoder_flops(dims):
    d, T = dims.n_audio_state, dims.n_audio_ctx
    conv = 2 * T * 2 * d * 3 * dims.n_mels + 2 * T * d * 3 * d
    layer = 4 * T * d * d * 2 + 2 * dims.n_audio_head * T * T * 64 * 2 + 2 * T * d * 4 * d * 2
    return conv + dims.n_audio_layer * layer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--tokens", type=int, default=145)
    ap.add_argument("--compute-type", default="float16", choices=["float16", "int8"], help="int8: decoder GEMV weights as int8 + row scales (config 5)")
    ap.add_argument("--cross-split", type=int, default=0, help="key splits of the decode cross-attention (1, 2, 4); 0 = 2")
    ap.add_argument("--fc2-tile-n", type=int, default=-1, help="-1 auto (16 with several passes in flight, else 8)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-dtw", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tokens", type=int, default=6)
    ap.add_argument("--streams", type=int, default=3, help="launch sequences in flight per GPU (one engine context + HIP stream + host thread each)")
    ap.add_argument("--coalesce", type=int, default=1,
                    help="16-chunk requests merged into one pass of the hot path (rows are independent: results are "
                         "identical, the decoder weights are streamed once per pass instead of once per request)")
    ap.add_argument("--no-extra", action="store_true", help="skip the additional coalesced-passes measurement (N=1 only)")
    ap.add_argument("--step-variant", type=int, default=1)
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI) or gloo (rehearsal on a 1-GPU box)")
    ap.add_argument("--host-input", action="store_true",
                    help="PCM starts in pinned host memory and is copied to HBM inside the timed region (the PCIe-inclusive rate; never `value`)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group even for WORLD_SIZE=1 (under torchrun)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--align", action="store_true", help="also run the wav2vec2-base CTC forward + forced alignment DP per chunk (config 4)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    use_dist = world > 1 or (args.force_dist and "RANK" in os.environ)   # --force-dist: rehearse the RCCL path with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.share_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    n_gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from whisperx_mlx_amd import weights
    from whisperx_mlx_amd.engine import WhisperHipEngine
    from whisperx_mlx_amd.tokenizer import get_tokenizer
    from tests.synth import speechlike_audio

    dims = weights.MODEL_DIMS[args.model]
    B = args.batch
    C = max(1, args.coalesce)
    extra = (world == 1 and not args.no_extra and C == 1 and B == 16)    # also measure 3 requests per pass, 2 passes in flight
    BE = B * max(C, 3 if extra else 1)            # most rows one pass of the hot path will carry
    ck = weights.random_checkpoint(dims, seed=0, std=0.02, device=dev)
    packed = weights.pack(ck, dims, dev)
    if args.compute_type == "int8":
        packed = weights.quantize_packed_decoder(packed, dims)
    heads = weights.default_alignment_heads(args.model, dims)
    engines = [WhisperHipEngine(dims, packed, max_batch=BE, device_index=local_rank, alignment_heads=heads)
               for _ in range(max(1, args.streams, 2 if extra else 1))]
    all_engines = engines
    engines = all_engines[: max(1, args.streams)]
    eng = engines[0]
    aligners = None
    if args.align:
        from whisperx_mlx_amd.w2v import W2VConfig, W2VHipModel, random_state_dict
        wcfg = W2VConfig()
        wsd = random_state_dict(wcfg, seed=1)
        from whisperx_mlx_amd.w2v import pack_w2v
        wpacked = pack_w2v(wsd, wcfg, dev)
        aligners = []
        for e in engines:
            m = W2VHipModel(wcfg, wpacked, device_index=local_rank)
            m.stream = e.stream
            aligners.append(m)
        g = torch.Generator().manual_seed(7)
        align_tok = torch.randint(1, wcfg.vocab, (BE, 400), generator=g, dtype=torch.int32).to(dev)
        align_N = torch.full((BE,), 400, dtype=torch.int32, device=dev)
    tok = get_tokenizer(dims.n_vocab)
    prompt = tok.sot_sequence()

    # synthetic 30-minute file, cut into 60 fixed 30 s chunks; rank r takes chunks r, r+N, ...
    audio = speechlike_audio(1800.0, seed=1234)
    chunks = audio.reshape(60, 480000)
    n_batches = args.warmup + args.steps
    pcm_batches = []
    for s in range(n_batches):
        idx = [((s * B + i) * n_gpus + rank) % 60 for i in range(B)]
        pcm_batches.append(torch.from_numpy(chunks[idx]).pin_memory() if args.host_input else torch.from_numpy(chunks[idx]).to(dev))
    n_valid_all = torch.full((BE,), 480000, dtype=torch.int32, device=dev)
    rec_w = dims.n_text_ctx + 4

    ev = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731
    host_ms = {"decode_enqueue": 0.0}

    split = {"v": 2, "fc2": 0}

    def one_step(pcm, e):
        """enqueues one whole pass (pcm: R = 16 x requests rows) on engine e's own stream (no host sync):
        with --streams > 1 consecutive passes run concurrently on the GPU and fill each other's launch gaps"""
        st = e.stream
        R = pcm.shape[0]
        n_valid = n_valid_all[:R]
        with torch.cuda.stream(st):
            marks = [ev() for _ in range(5)]
            marks[0].record(st)
            if args.host_input:
                pcm = pcm.to(dev, non_blocking=True)      # H2D over PCIe on the pass's own stream
            mel = e.logmel(pcm, n_valid)
            marks[1].record(st)
            enc = e.encode(mel)
            marks[2].record(st)
            h0 = time.perf_counter()
            out = e.decode(enc, tok, prompt, rules=0, forced_len=args.tokens, capture_qk=not args.no_dtw,
                           use_graph=not args.no_graph, cross_split=split["v"], step_variant=args.step_variant,
                           fc2_tile_n=split["fc2"])
            host_ms["decode_enqueue"] += (time.perf_counter() - h0) * 1e3
            marks[3].record(st)
            ws = e.dtw_launch(out, tok.eot) if not args.no_dtw else None
            if aligners is not None:
                al = aligners[engines.index(e)]
                logp, T = al.emissions_device(pcm, [480000] * R)
                al.ctc_align(logp, torch.tensor(T, dtype=torch.int32), align_tok[:R], align_N[:R], 0, 2)
            marks[4].record(st)
            rec = torch.zeros(R, rec_w, dtype=torch.int32, device=dev)
            rec[:, : dims.n_text_ctx] = out.tokens
        return rec, marks, ws

    def pass_pcm(steps_of_pass, base):
        return torch.cat([pcm_batches[base + s] for s in steps_of_pass]) if len(steps_of_pass) > 1 else pcm_batches[base + steps_of_pass[0]]

    def timed_run(C, engines):
        # key splits of the decode cross-attention: 2 at 16 rows; 48-row passes have blocks enough without a split
        split["v"] = args.cross_split if args.cross_split > 0 else (1 if C > 1 else 2)
        # several passes in flight: the K = 4d GEMV as 80 fat blocks (leaves CUs to the other passes); alone: 160 blocks
        split["fc2"] = args.fc2_tile_n if args.fc2_tile_n >= 0 else (16 if len(engines) > 1 else 0)
        # a pass takes up to C consecutive requests (steps); the last one of a run may be partial
        passes = [list(range(a, min(a + C, args.steps))) for a in range(0, args.steps, C)]
        stage_ms = {"logmel": 0.0, "encode": 0.0, "decode": 0.0, "dtw": 0.0}
        # warm-up: every engine once per pass size it will see (captures its hipGraphs), >= --warmup requests in all
        sizes = sorted({len(p) for p in passes}, reverse=True)
        for k, e in enumerate(engines):
            for n in sizes:
                one_step(torch.cat([pcm_batches[(k + i) % max(1, args.warmup)] for i in range(n)]) if n > 1 else pcm_batches[k % max(1, args.warmup)], e)
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        host_ms["decode_enqueue"] = 0.0
        t0 = time.perf_counter()
        recs, all_marks = [None] * len(passes), [None] * len(passes)

        def worker(k):
            # one host thread per engine context: kernel launches block when the HW queue is full, so
            # concurrent passes need concurrent launchers (ctypes drops the GIL inside libwxhip.so)
            torch.cuda.set_device(dev)
            for i in range(k, len(passes), len(engines)):
                rec, marks, _ = one_step(pass_pcm(passes[i], args.warmup), engines[k])
                recs[i], all_marks[i] = rec, marks

        if len(engines) == 1:
            worker(0)
        else:
            import threading
            th = [threading.Thread(target=worker, args=(k,)) for k in range(len(engines))]
            for t in th:
                t.start()
            for t in th:
                t.join()
        for e in engines:
            torch.cuda.current_stream(dev).wait_stream(e.stream)
        local = torch.cat(recs).reshape(args.steps, B, rec_w)
        if use_dist:
            if args.dist_backend == "nccl":
                gathered = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=dev)
                dist.all_gather_into_tensor(gathered, local)      # the one RCCL collective (xGMI)
            else:
                lc = local.cpu()
                gathered = torch.empty((world * lc.shape[0],) + tuple(lc.shape[1:]), dtype=lc.dtype)
                dist.all_gather_into_tensor(gathered, lc)
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        for marks in all_marks:
            for i, k in enumerate(("logmel", "encode", "decode", "dtw")):
                stage_ms[k] += marks[i].elapsed_time(marks[i + 1]) / args.steps
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        if use_dist:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

        return dt, stage_ms, passes

    dt, stage_ms, passes = timed_run(C, engines)
    main_split = split["v"]
    audio_s = n_gpus * args.steps * B * 30.0
    value = audio_s / dt
    result = {
        "metric": f"real-time factor (x) {args.model} batch={B}; word-timestamp path included",
        "value": round(value, 2), "unit": "x realtime (audio s / wall s)", "n_gpus": n_gpus,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16" if args.compute_type == "float16" else "f16 (int8 decoder GEMV weights)",
        "data": "synthetic 16 kHz audio (rng 1234), seeded random fp16 weights, forced 145 sampled tokens",
        "config": {"workload": f"whisper-{args.model} fp16 batch_size={B}, 30 min synthetic 16 kHz audio in 30 s chunks, "
                               f"log-mel + encoder + greedy decode ({args.tokens} tokens) + cross-attention DTW",
                   "global_batch": B * n_gpus, "chunks_per_step": B, "requests_coalesced_per_pass": C,
                   "passes_in_flight_per_gpu": len(engines), "batches_in_flight_per_gpu": len(engines) * C, "cross_split": main_split,
                   "parallelism": f"dp{n_gpus} (chunk shards, 1 RCCL all_gather)"},
        "per_gpu_rtf": round(value / n_gpus, 2),
        "stages_ms": {k: round(v, 3) for k, v in stage_ms.items()},
        "align_stage": bool(args.align), "input": "pinned host memory (PCIe copy timed)" if args.host_input else "resident in HBM",
        "host_enqueue_ms_per_step": round(host_ms["decode_enqueue"] / args.steps, 3),
    }

    if extra:
        # same K requests again, 3 merged per pass of the hot path and 2 passes in flight (rows are
        # independent, tests/test_gpu_whisper.py::test_greedy_decode_coalesced_requests): reported beside
        # `value`, which stays one request (16 chunks) per pass
        dt3, st3, _ = timed_run(3, all_engines[:2])
        result["coalesced_passes"] = {"value": round(args.steps * B * 30.0 / dt3, 2), "unit": result["unit"],
                                      "requests_per_pass": 3, "rows_per_pass": 3 * B, "passes_in_flight": 2, "cross_split": split["v"],
                                      "ms_per_step": round(dt3 / args.steps * 1e3, 3),
                                      "stages_ms": {k: round(v, 3) for k, v in st3.items()}}

    # one extra batch alone on the GPU (outside the timed region): uncontended per-stage times
    torch.cuda.synchronize(dev)
    split["v"], split["fc2"] = main_split, 0
    _rec, m1, _ = one_step(pass_pcm(passes[0], args.warmup), engines[0])
    torch.cuda.synchronize(dev)
    R1 = len(passes[0]) * B
    single_ms = {k: m1[i].elapsed_time(m1[i + 1]) for i, k in enumerate(("logmel", "encode", "decode", "dtw"))}
    result["stages_ms_single_stream"] = {k: round(v, 3) for k, v in single_ms.items()}
    result["stages_ms_single_stream"]["rows"] = R1
    result["single_stream_rtf"] = round(R1 * 30.0 / (sum(single_ms.values()) * 1e-3), 1)

    if rank == 0:
        # ---- roofline of the dominant kernel (decode cross-attention: streams every sequence's
        # cross K/V once per layer per step), timed live with HIP events on the engine's stream
        iters = dims.n_text_layer * 4
        ms = eng.probe(0, R1, iters, main_split)
        bytes_launch = algorithmic_bytes(dims, R1, "cross_attn")
        ach = bytes_launch / (ms * 1e-3) / 1e9
        traffic = None   # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside bench.py)
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if os.path.exists(pmc) and args.model == "large-v3":
            with open(pmc) as f:
                for k, v in json.load(f)["kernels"].items():
                    if "dec_cross_attn_kernel" in k and v.get("rows", 16) == R1:
                        traffic = v.get("hbm_bytes_per_launch_corrected")
        result["roofline"] = {"kernel": "dec_cross_attn_kernel", "bound": "hbm", "achieved": round(ach, 1),
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                              "traffic": traffic, "avg_launch_us": round(ms * 1e3, 2),
                              "algorithmic_bytes_per_launch": bytes_launch, "rows_per_launch": R1}
        # secondary figures: whole decode step against the HBM roof, encoder against the MFMA roof
        n_pos = len(prompt) + args.tokens - 1
        step_bytes = algorithmic_bytes(dims, R1, "decode_step", t_self=n_pos // 2)
        dec_gbs = step_bytes * n_pos / (single_ms["decode"] * 1e-3) / 1e9
        enc_tf = encoder_flops(dims) * R1 / (single_ms["encode"] * 1e-3) / 1e12
        fc1_ms = eng.probe(1, R1, 8)
        fc1_tf = 2.0 * R1 * 1500 * dims.n_audio_state * 4 * dims.n_audio_state / (fc1_ms * 1e-3) / 1e12
        att_ms = eng.probe(2, R1, 8)
        att_tf = 4.0 * R1 * dims.n_audio_head * 1500 * 1500 * 64 / (att_ms * 1e-3) / 1e12
        result["roofline_more"] = {
            "decode_loop_hbm": {"achieved_GBs": round(dec_gbs, 1), "frac": round(dec_gbs / HBM_PEAK_GBS, 4),
                                "bytes_per_step": step_bytes, "positions": n_pos},
            "encoder_mfma": {"achieved_TFLOPs": round(enc_tf, 1), "frac": round(enc_tf / MFMA_PEAK_TFLOPS, 4)},
            "enc_fc1_gemm": {"achieved_TFLOPs": round(fc1_tf, 1), "frac": round(fc1_tf / MFMA_PEAK_TFLOPS, 4), "ms": round(fc1_ms, 3)},
            "enc_attention": {"achieved_TFLOPs": round(att_tf, 1), "frac": round(att_tf / MFMA_PEAK_TFLOPS, 4), "ms": round(att_ms, 3)},
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args, dims, ck, chunks, tok, prompt)
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, dims, ck, chunks, tok, prompt):
    """The oracle (kind "port": our torch-CPU fp32 restatement) on the host cores, on a
    bounded sample: ONE 30 s chunk, batch 1, `cpu_tokens` forced tokens; the decode time is
    scaled to the workload's token count (every step costs the same on the CPU: it is
    weight-bandwidth bound)."""
    from oracle import decoding as OD, logmel as OL, whisper_ref as OW
    from whisperx_mlx_amd.audio import mel_filters
    threads = torch.get_num_threads()
    w = {k: v.float().cpu() for k, v in ck.items()}
    t0 = time.perf_counter()
    mel = OL.log_mel_chunks([chunks[0]], [480000], mel_filters(dims.n_mels))
    t1 = time.perf_counter()
    enc = OW.encoder_forward(w, dims, torch.from_numpy(mel))
    t2 = time.perf_counter()
    OD.greedy_decode(w, dims, enc, OD.Specials.for_vocab(dims.n_vocab), prompt, rules=0, forced_len=args.cpu_tokens)
    t3 = time.perf_counter()
    dec_full = (t3 - t2) * (len(prompt) + args.tokens - 1) / (len(prompt) + args.cpu_tokens - 1)
    wall = (t2 - t0) + dec_full
    return {"value": round(30.0 / wall, 3), "unit": "x realtime (audio s / wall s)", "cores": threads, "kind": "port",
            "host_cpus": os.cpu_count(),
            "sample": f"1 chunk (30 s), batch 1: log-mel {t1 - t0:.2f}s + encoder {t2 - t1:.2f}s + "
                      f"{args.cpu_tokens} decode steps {t3 - t2:.2f}s scaled to {args.tokens} tokens ({dec_full:.2f}s); torch-CPU fp32"}


if __name__ == "__main__":
    main()
