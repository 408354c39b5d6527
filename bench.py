#!/usr/bin/env python3
"""bench.py -- real-time factor of the HIP hot path on MI355X (BASELINE.json metric), measured THROUGH the drop-in API.

A "step" = one request of B = 16 synthetic 30 s chunks, already resident in HBM, through
`WhisperHipBackend.transcribe_batch` (the seam the reference calls at whisperx/asr.py:80-87): log-mel -> Whisper
encoder -> cross-KV projection -> greedy decode (device-resident loop, DecodingOptions-default logit filters on,
forced token count) -> cross-attention DTW -> token ids, log-probabilities and word times copied out and assembled
into the reference's result dict.  The timed region is ONE transcribe_batch call over K requests: the backend's own
scheduler keeps 4 passes in flight (engine contexts + launcher threads on streams it has checked to run side by side).  Weights are seeded N(0, 0.02^2) fp16 in the
exact large-v3 shapes (no checkpoint ships with the reference), so the decode length is forced to the reference's
measured mean (145 sampled + 3 prompt tokens, BASELINE.md, tests/golden/gold30m_windows.json).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Chunks shard over ranks with no data-path collective; ONE RCCL all_gather of the fixed-size result records
(tokens, log-probabilities, word spans) closes the timed region.  Rank 0 prints ONE JSON line.

  --ckpt-dir DIR --audio FILE   real weights + real audio when a box has them: token parity against the oracle on a
                                bounded sample, mean token count, and word_mae_ms against the gold standard
                                (tests/golden/gold30m/30m.json.gz).  Without them "word_mae_ms" is null.
"""
import argparse
import csv
import glob
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


def host_cpu_share():
    """CPUs this process may really use: the cgroup's quota when there is one (a 1-GPU box of the pool: 16 of the host's 256),
    else the affinity mask.  torch's default thread count follows the host's CPU count, which oversubscribes a box with a
    quota eight-fold (fp32 matmul: 1.4 TFLOP/s on 16 threads, 0.6 on 128)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def algorithmic_bytes(dims, B, kind, t_self=75):
    """fp16 bytes one launch must move."""
    d = dims.n_text_state
    if kind == "cross_attn":        # one layer: K and V of every sequence, read once
        return B * 2 * dims.n_audio_ctx * d * 2
    if kind == "cq_cross_attn":     # the fused launch: + the cross-Q weight [d][d], read once
        return B * 2 * dims.n_audio_ctx * d * 2 + 2 * d * d
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


def committed_profile(kernel_substr, pattern="r0[3-9]_bench_kernel_stats.csv", rows=None):
    """in-situ average duration (us) of a kernel from the newest committed rocprofv3 --kernel-trace summary of this same
    command.  With `rows` (round 4 onwards): the row of that launch width in profiles/rNN_bench_kernel_stats_by_width.csv
    (tools/trace_overlap.py groups the trace's dispatches by kernel and grid size -- one row per launch width, so the
    16-row launches of the command's batch-16 phase no longer dilute the wide launches' average); without, or when no
    by-width file is committed, the per-kernel row of rNN_bench_kernel_stats.csv.  None when no profile is committed."""
    if rows is not None:
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[4-9]_bench_kernel_stats_by_width.csv")))
        if files:
            with open(files[-1]) as f:
                for row in csv.DictReader(f):
                    if kernel_substr in row.get("Name", "") and int(float(row.get("Rows", -1))) == int(rows):
                        # inside the part of the trace where the passes are in flight together (what the live timer
                        # measures); the whole-trace average also covers warm-up, ramps and tails, where a launch has
                        # fewer competitors for the HBM
                        ns = row.get("InFlightAverageNs") or row["AverageNs"]
                        return float(ns) / 1e3, os.path.relpath(files[-1], ROOT) + f" (row: {rows} rows, in-flight window; whole trace {float(row['AverageNs']) / 1e3:.1f} us)"
            return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None, None
    with open(files[-1]) as f:
        for row in csv.DictReader(f):
            if kernel_substr in row.get("Name", ""):
                return float(row["AverageNs"]) / 1e3, os.path.relpath(files[-1], ROOT)
    return None, None


def committed_plan():
    """[sorted rows of the passes, passes in flight] of the run the committed trace (profiles/rNN_bench_overlap.txt) is of"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]_bench_plan.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        d = json.load(f)
    return [sorted(int(r) for r in d["rows"]), int(d["passes_in_flight"])]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: this process becomes the launcher.  It starts
    N fresh rank processes (`python -m torch.distributed.run`, rendezvous on 127.0.0.1) BEFORE anything has initialised
    the GPU -- children, never a re-exec -- waits for them and exits with their code; rank 0 of the children prints the
    JSON line.  `torch.cuda.device_count()` does not initialise the GPU on this image."""
    import socket
    import subprocess
    if args.share_gpu and args.dist_backend == "nccl":
        print("bench.py: --share-gpu is a rehearsal on a 1-GPU box and needs --dist-backend gloo (RCCL refuses two ranks on one "
              "device: 'Duplicate GPU detected')", file=sys.stderr)
        sys.exit(2)
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not args.share_gpu and args.dist_backend == "nccl":
        print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) are visible", file=sys.stderr)
        sys.exit(3)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(sys.argv[0]), *argv]
    print(f"bench.py: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    sys.exit(subprocess.run(cmd, env=env).returncode)


def inflight_window(kernel_substr, bytes_launch, layer_bytes_extra=0):
    """the dominant kernel inside the part of the committed trace where all passes are in flight (tools/trace_overlap.py
    report of the same command, profiles/rNN_bench_overlap.txt): its average there, and how many kernels run at once"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]_bench_overlap.txt")))
    if not files:
        return None
    us = conc = share = launches = window_ms = None
    for line in open(files[-1]):
        f = line.split()
        if len(f) >= 4 and f[0] == "dec_cq_xattn" and kernel_substr.startswith("dec_cq_xattn"):
            launches, us, share = int(f[1]), float(f[2]), float(f[3])
        if "kernels running at once" in line:
            conc = float(line.rsplit(":", 1)[1])
        if line.startswith("window:"):
            window_ms = float(f[1])
    if us is None:
        return None
    gbs = bytes_launch / (us * 1e-6) / 1e9
    out = {"us": us, "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "kernels_running_at_once": conc,
           "source": os.path.relpath(files[-1], ROOT),
           "note": "average launch while three passes are in flight: a launch shares the HBM with the other passes' launches"}
    if conc and share and launches and window_ms:
        # what the launches of this kernel that run side by side reach TOGETHER, and every decode kernel of the window:
        # a layer of a pass = this launch + the other 13 d^2 weights of the layer (layer_bytes_extra), streamed once per pass
        agg = launches * bytes_launch / (window_ms * 1e-3) / 1e9
        out["launches_of_this_kernel_at_once"] = round(share * conc, 2)
        out["aggregate"] = {"achieved": round(agg, 1), "frac": round(agg / HBM_PEAK_GBS, 4),
                            "note": "bytes of all launches of this kernel in the window / the window: the launches in flight together"}
        if layer_bytes_extra:
            dec = launches * (bytes_launch + layer_bytes_extra) / (window_ms * 1e-3) / 1e9
            out["decode_window_hbm"] = {"achieved": round(dec, 1), "frac": round(dec / HBM_PEAK_GBS, 4),
                                        "note": "algorithmic bytes of every decode layer run in the window (this launch + the layer's other GEMV weights) / the window"}
    return out


def main(argv=None, make_backend=None):
    """make_backend: tests only (tests/test_bench_ranks.py drives the rank launch, the process group and the gather on
    the CPU with a stand-in backend); the bench itself always builds a WhisperHipBackend and needs a GPU."""
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--tokens", type=int, default=145)
    ap.add_argument("--compute-type", default="float16", choices=["float16", "int8"], help="int8: decoder GEMV weights as int8 + row scales (config 5)")
    ap.add_argument("--streams", type=int, default=0, help="passes in flight per GPU (engine contexts of the backend's scheduler); 0 = the backend's own choice")
    ap.add_argument("--rows-per-pass", type=int, default=0, help="rows per pass of the hot path; 0 = the backend's own plan for the job (requests merged into passes of up to 128 rows)")
    ap.add_argument("--coalesce", type=int, default=0, help="requests of --batch chunks a pass may hold; 0 = the backend's default (contexts of 128 rows, planned per job)")
    ap.add_argument("--rules", type=int, default=127, help="logit-filter rule bits (127 = DecodingOptions defaults, mlx_lightning.py:187-193)")
    ap.add_argument("--no-dtw", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tokens", type=int, default=16, help="real decode steps of the CPU baseline (batch 16)")
    ap.add_argument("--no-extra", action="store_true", help="skip the additional coalesced-passes measurement (N=1 only)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI) or gloo (rehearsal on a 1-GPU box)")
    ap.add_argument("--host-input", action="store_true",
                    help="PCM handed over as host numpy arrays: staged through pinned memory and copied to HBM inside the timed region (the PCIe-inclusive rate; never `value`)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group even for WORLD_SIZE=1 (under torchrun)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--no-align", action="store_true", help="skip the wav2vec2-base CTC forward + forced-alignment DP measurement (config 4's second model, reported as align_stage)")
    ap.add_argument("--longform", type=float, default=0.0, metavar="HOURS",
                    help="config 5: HOURS of synthetic long-form audio through batch_processor.batch_transcribe instead of the 16-chunk requests")
    ap.add_argument("--ckpt-dir", default=os.environ.get("WX_CKPT_DIR"), help="real Whisper checkpoint directory (config.json + safetensors)")
    ap.add_argument("--audio", default=os.environ.get("WX_AUDIO_NPY"), help="real 16 kHz mono audio (.npy / .wav) matching the gold standard")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args, argv)            # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` or as "
              f"`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`", file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    use_dist = world > 1 or (args.force_dist and "RANK" in os.environ)   # --force-dist: rehearse the RCCL path with one rank
    if use_dist:
        # the process group comes first: nothing has touched the GPU yet (and nothing re-execs after this point)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.share_gpu:
            local_rank = 0
        if args.dist_backend == "nccl":
            if torch.cuda.device_count() <= local_rank:
                print(f"bench.py: rank {rank} has no GPU {local_rank} ({torch.cuda.device_count()} visible)", file=sys.stderr)
                sys.exit(3)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
        if dist.get_world_size() != args.gpus:
            print(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}", file=sys.stderr)
            sys.exit(2)
    n_gpus = world
    on_gpu = make_backend is None
    if on_gpu:
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
        dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(dev)
    else:
        dev = torch.device("cpu")
    sync = (lambda: torch.cuda.synchronize(dev)) if on_gpu else (lambda: None)
    # what the process group itself saw: every rank's (rank, device) through the collective backend
    ranks_seen = [[rank, local_rank]]
    if use_dist:
        mine = torch.tensor([rank, local_rank], dtype=torch.int32, device=dev if args.dist_backend == "nccl" else "cpu")
        seen = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(seen, mine)
        ranks_seen = [[int(v) for v in t.tolist()] for t in seen]
        assert sorted(r for r, _ in ranks_seen) == list(range(args.gpus)), ranks_seen

    from whisperx_mlx_amd import parallel as PAR
    from whisperx_mlx_amd import weights
    from whisperx_mlx_amd.synth import speechlike_audio

    from whisperx_mlx_amd import HW_QUEUES
    B = args.batch
    extra = (on_gpu and world == 1 and not args.no_extra and B == 16 and not args.longform)    # also measure one 16-chunk request per pass, four passes in flight (round 2's `value`)
    real = bool(args.ckpt_dir)
    if on_gpu:
        from whisperx_mlx_amd.backend import WhisperHipBackend
        make_backend = WhisperHipBackend
    be = make_backend(args.ckpt_dir if real else args.model, device="cuda", device_index=local_rank,
                      compute_type=args.compute_type, max_batch=B, coalesce=args.coalesce or None,
                      random_init=not real, seed=0, passes_in_flight=args.streams or None, rules=args.rules,
                      max_rows=(48 if args.share_gpu else None))     # --share-gpu: several ranks' contexts in ONE GPU's memory
    # rows per pass and passes in flight: the backend's own plan for the job (backend.plan_passes: requests of B chunks
    # merged into passes of up to 128 rows, three in flight) unless --rows-per-pass / --streams pin them
    rows_arg = args.rows_per_pass or None
    n_streams = args.streams if args.streams > 0 else None
    dims = be.dims
    eng = be.engine
    tok = be.tokenizer
    prompt = tok.sot_sequence("en", "transcribe")
    forced = 0 if real else args.tokens
    wt = False if args.no_dtw else "dtw"

    if args.longform:
        result = longform(args, be, dims, n_gpus, rank, use_dist, dist, dev)
        if rank == 0:
            print(json.dumps(result), flush=True)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    # synthetic 30-minute file, cut into 60 fixed 30 s chunks; rank r takes chunks r, r+N, ...
    audio = speechlike_audio(1800.0, seed=1234)
    chunks = audio.reshape(60, 480000)
    chunks_dev = None if args.host_input else torch.from_numpy(chunks).to(dev)

    def request_segments(first_step, n_steps):
        """the 16-chunk requests first_step .. first_step + n_steps - 1 as transcribe_batch segments (asr.py:70-73)"""
        segs = []
        for s in range(first_step, first_step + n_steps):
            for i in range(B):
                j = ((s * B + i) * n_gpus + rank) % 60
                segs.append({"start": 30.0 * j, "end": 30.0 * (j + 1), "audio": chunks[j] if args.host_input else chunks_dev[j]})
        return segs

    def run(first_step, n_steps, rows_per_pass, in_flight):
        return be.transcribe_batch(request_segments(first_step, n_steps), batch_size=B, language="en", word_timestamps=wt,
                                   forced_len=forced, rows_per_pass=rows_per_pass, passes_in_flight=in_flight,
                                   return_chunks=True)

    tail_ms = {}        # host tail of the last timed run: record packing, the collective + table

    def timed_run(rows_per_pass, in_flight):
        # warm-up: --warmup requests through the same call, and never fewer than the timed call has: the backend plans the
        # passes from the size of the job, and every context must have captured the hipGraphs of that plan's launch shape
        run(0, max(args.warmup, args.steps), rows_per_pass, in_flight)
        be.stage_ms = {}
        sync()
        if use_dist:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        res = run(args.warmup, args.steps, rows_per_pass, in_flight)
        t1 = time.perf_counter()
        recs = PAR.pack_records(res["chunks"], [c["segment"] * n_gpus + rank for c in res["chunks"]], align=False)
        t2 = time.perf_counter()
        # the one collective (RCCL over xGMI); what comes back is the job's RecordTable: one int32 array ordered by chunk
        # id, a record becomes a dict when it is read (parallel.RecordTable)
        gathered = PAR.gather_records(recs, counts=[recs.shape[0]] * world) if use_dist else None
        t3 = time.perf_counter()
        sync()
        if use_dist:
            dist.barrier()
        sync()
        dt = time.perf_counter() - t0
        tail_ms.update(pack=round((t2 - t1) * 1e3, 2), gather_and_table=round((t3 - t2) * 1e3, 2))
        if use_dist:
            assert len(gathered) == world * recs.shape[0]
            assert gathered.chunk_ids.tolist() == list(range(world * recs.shape[0]))
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        stage_ms = {k: v / args.steps for k, v in be.stage_ms.items()}
        be.stage_ms = None
        return dt, stage_ms, res

    def live_launch_timer(rows_per_pass, in_flight):
        """the same K requests once more with the launch timer on (wx_decode_opts.profile_launches): the fused decode
        launches are timed on the device, on whatever stream and hipGraph they run in -- the LIVE duration behind a
        `roofline` block"""
        if not (on_gpu and hasattr(be, "profile_launches")):
            return None
        be.profile_launches = True
        try:
            run(0, max(args.warmup, args.steps), rows_per_pass, in_flight)            # captures the timed graphs
            for e in be.engines:
                e.launch_profile()
            dt_t, _st, _res = timed_run(rows_per_pass, in_flight)
            recs = [e.launch_profile() for e in be.engines]
            n_l = sum(r[1] for r in recs)
            if n_l:
                return {"us": sum(r[0] * r[1] for r in recs) / n_l, "launches": n_l,
                        "value_with_timer_on": round(audio_s / dt_t, 2)}
        finally:
            be.profile_launches = False
        return None

    dt, stage_ms, res = timed_run(rows_arg, n_streams)
    plan = dict(getattr(be, "last_plan", None) or {"rows": [B], "launch_rows": B, "passes_in_flight": n_streams or 1})
    n_chunks = len(res["chunks"])
    assert n_chunks == args.steps * B
    audio_s = n_gpus * args.steps * B * 30.0
    value = audio_s / dt
    n_tok = float(np.mean([len(c["tokens"]) for c in res["chunks"]]))
    n_text = float(np.mean([sum(t < tok.eot for t in c["tokens"]) for c in res["chunks"]]))
    n_words = float(np.mean([len(c.get("words", [])) for c in res["chunks"]]))
    result = {
        "metric": f"real-time factor (x) {args.model} batch={B}; word-timestamp path included",
        "value": round(value, 2), "unit": "x realtime (audio s / wall s)", "n_gpus": n_gpus,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16" if args.compute_type == "float16" else "f16 (int8 decoder GEMV weights)",
        "data": ("real checkpoint + audio" if real else
                 f"synthetic 16 kHz audio (rng 1234), seeded random fp16 weights, forced {args.tokens} sampled tokens"),
        "config": {"workload": f"whisper-{args.model} fp16 batch_size={B}, {args.steps * B} chunks of 30 s per GPU (the 60 chunks of the 30 min "
                               f"synthetic 16 kHz file cycled: {args.steps * B / 2:g} min of audio) in ONE WhisperHipBackend.transcribe_batch call: "
                               f"log-mel + encoder + greedy decode ({args.tokens} tokens, "
                               f"logit filters rules={args.rules}) + cross-attention DTW + result dicts",
                   "global_batch": B * n_gpus, "chunks_per_step": B,
                   "rows_per_pass": plan["rows"] if len(set(plan["rows"])) > 1 else plan["rows"][0], "launch_rows": plan["launch_rows"],
                   "requests_per_pass": round(max(plan["rows"]) / B, 2),
                   "passes_in_flight_per_gpu": plan["passes_in_flight"], "hw_queues": HW_QUEUES,
                   "parallelism": f"dp{n_gpus} (chunk shards, 1 RCCL all_gather)",
                   "ranks_seen_by_the_process_group": ranks_seen, "dist_backend": args.dist_backend if use_dist else None},
        "per_gpu_rtf": round(value / n_gpus, 2),
        "stages_ms": {k: round(v, 3) for k, v in stage_ms.items()},
        "mean_sampled_tokens": round(n_tok, 1), "mean_text_tokens": round(n_text, 1), "mean_dtw_words": round(n_words, 1),
        "word_mae_ms": None,
        "input": "host arrays (pinned staging + PCIe copy timed)" if args.host_input else "resident in HBM",
        "gather_tail_ms": dict(tail_ms),
    }

    if on_gpu:
        free_b, total_b = torch.cuda.mem_get_info(dev)
        result["gpu_memory_gb"] = {"in_use_after_the_timed_run": round((total_b - free_b) / 2 ** 30, 1), "total": round(total_b / 2 ** 30, 1),
                                   "contexts": len(getattr(be, "engines", [])), "rows_per_context": getattr(eng, "max_batch", None),
                                   "note": "model weights + their tile-blocked decode copies + ONE encoder workspace + the engine contexts "
                                           "(cross K/V, self-attention cache of 232 positions, alignment scores) + torch's cached blocks"}
    live_launch = live_launch_timer(rows_arg, n_streams)

    live16 = plan16 = None
    if extra:
        # BASELINE.json's metric says batch = 16: the same K requests again with one 16-chunk request per pass of the hot
        # path (the device batch the metric names), four passes in flight -- round 2's scheduling -- on the same build;
        # rows are independent, the tokens must be the same.  Its own roofline block follows below.
        lanes16 = be._default_lanes(B)
        dt1, st1, res1 = timed_run(B, lanes16)
        plan16 = dict(be.last_plan)
        same = [a["tokens"] == b["tokens"] for a, b in zip(res["chunks"], res1["chunks"])]
        result["value_batch16"] = {"value": round(args.steps * B * 30.0 / dt1, 2), "unit": result["unit"],
                                   "rows_per_pass": B, "passes_in_flight": plan16["passes_in_flight"],
                                   "ms_per_step": round(dt1 / args.steps * 1e3, 3),
                                   "stages_ms": {k: round(v, 3) for k, v in st1.items()},
                                   "tokens_identical_to_value_run": bool(all(same)),
                                   "note": "every pass of the hot path is ONE request of batch_size = 16 chunks (the device batch "
                                           "BASELINE.json's metric names); `value` lets the backend's scheduler merge requests into wider passes"}
        live16 = live_launch_timer(B, lanes16)

    if extra and not real and args.model == "large-v3":
        result["config2"] = config2(chunks_dev if chunks_dev is not None else torch.from_numpy(chunks).to(dev), wt, args.tokens, dev)
    if extra:
        result["job_30min"] = job_30min(be, chunks, chunks_dev, wt, forced, B, dev)
        result["vad_mix"] = vad_mix(be, audio, wt, B, dev)
        if not args.no_align:
            result["config4"] = config4(be, audio, B, dev)

    if on_gpu and use_dist and not args.no_align and not args.longform and not real:
        # N > 1: config 4 as written -- ASR + forced alignment sharded over the ranks, one gather
        result["config4"] = config4_sharded(be, audio, B, dev, n_gpus, rank, dist, args.dist_backend)

    if not on_gpu:                 # tests/test_bench_ranks.py: the rank launch, process group and gather are what is exercised
        if rank == 0:
            print(json.dumps(result), flush=True)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    # one extra request alone on the GPU (outside the timed region): uncontended per-stage times
    torch.cuda.synchronize(dev)
    be.stage_ms = {}
    be.transcribe_batch(request_segments(0, 1), batch_size=B, language="en", word_timestamps=wt, forced_len=forced,
                        rows_per_pass=B, passes_in_flight=1)
    single_ms = dict(be.stage_ms)
    be.stage_ms = None
    result["stages_ms_single_stream"] = {k: round(v, 3) for k, v in single_ms.items()}
    result["stages_ms_single_stream"]["rows"] = B
    result["single_stream_rtf"] = round(B * 30.0 / (sum(single_ms.values()) * 1e-3), 1)

    if not args.no_align and rank == 0 and n_gpus == 1:
        result["align_stage"] = align_stage(be, chunks_dev if chunks_dev is not None else torch.from_numpy(chunks).to(dev), B, dev)
    else:
        result["align_stage"] = False

    if real and args.audio and rank == 0:
        result.update(real_run(args, be))

    if rank == 0:
        # ---- roofline of the dominant kernel (decode cross-attention: streams every sequence's cross K/V once per
        # layer per step).  Live: HIP events on the engine's own stream around back-to-back launches rotating over the
        # layers (cold bytes every launch).  In situ: the same kernel's average duration inside the decode step chain,
        # from the committed rocprofv3 --kernel-trace --stats summary of this command; `frac` uses the in-situ duration
        # when a profile is committed (it is the longer of the two), else the live one.
        iters = dims.n_text_layer * 4
        KERNEL = "dec_cq_xattn_kernel"          # [LN + cross-Q GEMV] -> [cross attention] in one launch (csrc/declayer.hip)

        def roofline_of(plan, live, committed):
            """the roofline block of the dominant kernel for one scheduling of the job (`plan`: rows per pass, passes in
            flight; `live`: the launch timer's record of that run; `committed`: look the same plan up in profiles/)"""
            # rows that carry a chunk in one launch of the timed run (padding rows of a launch are not streamed)
            # A plan may mix pass widths (320 chunks: 112 + 112 + 96 rows).  Every pass issues the same number of launches, so
            # the timed region's launches carry the MEAN of the plan's rows: `achieved` = bytes of that mean launch over the
            # live average duration (= all bytes of the kernel over all of its launch time).  The launch probed alone, and
            # looked up in the PMC passes, is the widest one (the majority of the rows).
            rows_mean = float(np.mean(plan["rows"]))
            rows_launch = max(1, min(int(max(plan["rows"])), eng.max_batch))
            ms = eng.probe(13, rows_launch, iters)
            bytes_probe = algorithmic_bytes(dims, rows_launch, "cq_cross_attn")
            bytes_launch = int(round(algorithmic_bytes(dims, 0, "cq_cross_attn") + rows_mean * algorithmic_bytes(dims, 1, "cross_attn")))
            live_us = ms * 1e3
            # in situ: the average of the same kernel AT THIS LAUNCH WIDTH in the committed rocprofv3 kernel trace of THIS
            # command (profiles/rNN_bench_kernel_stats_by_width.csv: one row per kernel and grid size)
            situ_us, situ_src = committed_profile(KERNEL, rows=rows_launch) if (committed and args.model == "large-v3") else (None, None)
            # the duration behind `achieved`: measured live by the launch timer over the timed region of this run; the
            # committed rocprofv3 average of the same command stands beside it
            use_us = live["us"] if live else (situ_us if situ_us else live_us)
            ach = bytes_launch / (use_us * 1e-6) / 1e9
            traffic, traffic_src = None, None   # HBM bytes per launch from committed PMC passes (rocprofv3 cannot run inside bench.py)
            pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary.json")))
            if pmcs and args.model == "large-v3":
                with open(pmcs[-1]) as f:
                    meas = {v.get("rows", 16): v.get("hbm_bytes_per_launch_corrected") for k, v in json.load(f)["kernels"].items() if KERNEL in k}
                if meas:
                    # measured at 16 / 64 / 128 rows: the width nearest to the plan's mean launch, scaled by the algorithmic
                    # bytes (the launch moves 1.005-1.008 x its algorithmic bytes at every measured width)
                    r_m = min(meas, key=lambda r: abs(r - rows_mean))
                    b_m = algorithmic_bytes(dims, r_m, "cq_cross_attn")
                    traffic = int(round(meas[r_m] * bytes_launch / b_m))
                    traffic_src = os.path.relpath(pmcs[-1], ROOT) + (f", measured at {r_m} rows ({meas[r_m] / b_m:.4f} x algorithmic), scaled to the plan's mean launch" if b_m != bytes_launch else "")
            same_plan = committed and situ_us and committed_plan() == [sorted(int(r) for r in plan["rows"]), int(plan["passes_in_flight"])] and n_gpus == 1
            return {"kernel": KERNEL, "bound": "hbm", "achieved": round(ach, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_source": f"committed PMC passes ({traffic_src})" if traffic_src else None,
                    "avg_launch_us": round(use_us, 2),
                    "duration_source": (f"live: device launch timer over the timed region, {live['launches']} launches on the "
                                        f"{plan['passes_in_flight']} streams of the run (timer on costs: value {live['value_with_timer_on']}x)"
                                        if live else
                                        f"in situ with {plan['passes_in_flight']} passes in flight, {situ_src}" if situ_us
                                        else "live HIP-event probe, launches back to back on one stream (no committed profile of this command)"),
                    "in_situ_source": situ_src,
                    "live_probe_us": round(live_us, 2), "in_situ_us": round(situ_us, 2) if situ_us else None,
                    "alone": {"us": round(live_us, 2), "rows": rows_launch, "achieved": round(bytes_probe / (live_us * 1e-6) / 1e9, 1),
                              "frac": round(bytes_probe / (live_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                              "note": "the plan's widest launch with the GPU to itself (live probe)"},
                    "in_flight": (inflight_window(KERNEL, bytes_launch, 13 * dims.n_text_state * dims.n_text_state * 2) if same_plan else None),   # the committed trace is of THAT plan
                    "algorithmic_bytes_per_launch": bytes_launch, "rows_per_launch": round(rows_mean, 2), "rows_of_the_plan": [int(r) for r in plan["rows"]]}

        result["roofline"] = roofline_of(plan, live_launch, committed=not rows_arg and not n_streams)
        if plan16 is not None:
            # the same kernel at the metric's own device batch: 16 rows per launch (126.2 MB), four passes in flight
            result["value_batch16"]["roofline"] = roofline_of(plan16, live16, committed=True)
        # secondary figures: whole decode step against the HBM roof, encoder against the MFMA roof
        n_pos = len(prompt) + args.tokens - 1
        step_bytes = algorithmic_bytes(dims, B, "decode_step", t_self=n_pos // 2)
        dec_gbs = step_bytes * n_pos / (single_ms["decode"] * 1e-3) / 1e9
        enc_tf = encoder_flops(dims) * B / (single_ms["encode"] * 1e-3) / 1e12
        fc1_ms = eng.probe(1, B, 8)
        fc1_tf = 2.0 * B * 1500 * dims.n_audio_state * 4 * dims.n_audio_state / (fc1_ms * 1e-3) / 1e12
        att_ms = eng.probe(2, B, 8)
        att_tf = 4.0 * B * dims.n_audio_head * 1500 * 1500 * 64 / (att_ms * 1e-3) / 1e12
        # the whole step of the timed run against the HBM roof: algorithmic decode bytes of one 16-chunk step under the
        # plan's pass width (decoder weights once per pass position, shared by the pass's rows) over ms_per_step
        # (encoder, DTW and host time included in the denominator)
        rows_mean = float(np.mean(plan["rows"]))
        d_ = dims.n_text_state
        w_dec = 2 * (dims.n_text_layer * 14 * d_ * d_ + dims.n_vocab * d_)
        kv_row = dims.n_text_layer * 2 * dims.n_audio_ctx * d_ * 2 + dims.n_text_layer * 2 * (n_pos // 2) * d_ * 2
        step_bytes_plan = n_pos * (w_dec * B / rows_mean + B * kv_row)
        whole_gbs = step_bytes_plan / (result["ms_per_step"] * 1e-3) / 1e9
        result["roofline_more"] = {
            "whole_step_hbm": {"achieved_GBs": round(whole_gbs, 1), "frac": round(whole_gbs / HBM_PEAK_GBS, 4),
                               "algorithmic_bytes_per_step": int(step_bytes_plan), "rows_per_pass_mean": round(rows_mean, 1),
                               "note": "decode bytes of one 16-chunk step / ms_per_step of the timed run (encoder time included)"},
            "decode_loop_hbm": {"achieved_GBs": round(dec_gbs, 1), "frac": round(dec_gbs / HBM_PEAK_GBS, 4),
                                "bytes_per_step": step_bytes, "positions": n_pos},
            "encoder_mfma": {"achieved_TFLOPs": round(enc_tf, 1), "frac": round(enc_tf / MFMA_PEAK_TFLOPS, 4)},
            "enc_fc1_gemm": {"achieved_TFLOPs": round(fc1_tf, 1), "frac": round(fc1_tf / MFMA_PEAK_TFLOPS, 4), "ms": round(fc1_ms, 3)},
            "enc_attention": {"achieved_TFLOPs": round(att_tf, 1), "frac": round(att_tf / MFMA_PEAK_TFLOPS, 4), "ms": round(att_ms, 3)},
        }
        result["fused_launch_selfq_blocks"] = int(getattr(be, "selfq_blocks", 0))     # attention blocks that computed their query themselves (declayer.hip)
        if n_gpus == 1 and not args.no_cpu_baseline and not real:
            result["cpu_baseline"] = cpu_baseline(args, dims, eng.packed, chunks, prompt, be.suppress)
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def _vad_segments(audio_dev):
    """the synthetic 30-minute file cut where the reference's own run cut its 30-minute file (81 windows of 1.2 .. 30 s,
    tests/golden/gold30m_windows.json) + the token count the reference produced for every window (4 .. 199, mean 108)"""
    with open(os.path.join(ROOT, "tests", "golden", "gold30m_windows.json")) as f:
        wins = json.load(f)["windows"]
    starts = [w["start"] for w in wins]
    ends = [min(a + 30.0, b) for a, b in zip(starts, starts[1:] + [1800.0])]
    lens = [min(len(w["tokens"]), 224) for w in wins]
    segs = [{"start": a, "end": b, "audio": audio_dev[int(a * 16000): int(b * 16000)]} for a, b in zip(starts, ends)]
    return segs, lens, sum(b - a for a, b in zip(starts, ends))


def _best_of(fn, dev, n=3):
    """wall time of fn(): the median of n runs after one warm-up (these jobs take ~1 s: a single run is at the mercy of the
    host's scheduler)"""
    res = fn()                                     # graphs of the job's launch shapes
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        res = fn()
        torch.cuda.synchronize(dev)
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2], res, ts


def job_30min(be, chunks, chunks_dev, wt, forced, B, dev):
    """config 3's own job size: the 30-minute file as exactly 60 fixed 30 s chunks in ONE transcribe_batch call (what
    model.transcribe(audio_30min) does), resident in HBM and -- what a caller holding numpy arrays gets -- with the input
    on the host (pinned staging + PCIe copy inside the timed region).  The driver's `value` runs 320 chunks per call; a
    60-chunk job is ONE round of <= 16-row passes on four contexts (scheduler.plan_passes: 49..64 chunks -> 4 x 15 rows;
    the plan actually run is reported as `rows_per_pass`) and the host's share (text, words, records) weighs more."""
    out = {"chunks": 60, "audio_s": 1800.0, "unit": "x realtime (audio s / wall s)"}
    for name, src in (("resident", chunks_dev), ("host_input", chunks)):
        if src is None:
            continue
        segs = [{"start": 30.0 * j, "end": 30.0 * (j + 1), "audio": src[j]} for j in range(60)]
        dt, res, ts = _best_of(lambda: be.transcribe_batch(segs, batch_size=B, language="en", word_timestamps=wt, forced_len=forced,
                                                           return_chunks=True), dev)
        assert len(res["chunks"]) == 60
        out[name] = {"value": round(1800.0 / dt, 2), "wall_ms": round(dt * 1e3, 1), "runs_ms": [round(t * 1e3, 1) for t in ts],
                     "rows_per_pass": be.last_plan["rows"], "passes_in_flight": be.last_plan["passes_in_flight"]}
    return out


def config2(chunks_dev, wt, tokens, dev):
    """BASELINE.json config 2: whisper-tiny fp16, batch_size = 8, the 30-minute synthetic file as 60 chunks of 30 s in ONE
    transcribe_batch call (model map /root/reference/whisperx/backends/mlx_lightning.py:49-69; seeded weights, the same forced
    token count as `value`).  Its roofline is the encoder attention's: SURVEY 8a row 3 -- at d = 384 the 1500 x 1500 scores
    and PV products of six heads (13.8 GFLOP per chunk) outweigh the projections (7.1) and match the FFN (14.2), so
    `attn_full_kernel` is the largest single kernel of a chunk, MFMA-bound."""
    from whisperx_mlx_amd.backend import WhisperHipBackend
    B2 = 8
    be2 = WhisperHipBackend("tiny", device="cuda", device_index=dev.index or 0, max_batch=B2, random_init=True, seed=0)
    d2 = be2.dims
    segs = [{"start": 30.0 * j, "end": 30.0 * (j + 1), "audio": chunks_dev[j]} for j in range(60)]
    kw = dict(batch_size=B2, language="en", word_timestamps=wt, forced_len=tokens, return_chunks=True)
    dt, res, ts = _best_of(lambda: be2.transcribe_batch(segs, **kw), dev)
    assert len(res["chunks"]) == 60 and all(len(c["tokens"]) == tokens for c in res["chunks"])
    plan = dict(be2.last_plan)
    be2.stage_ms = {}
    be2.transcribe_batch(segs, **kw)
    torch.cuda.synchronize(dev)
    stages = {k: round(v, 3) for k, v in be2.stage_ms.items()}       # summed over the job's passes (GPU time on each pass's own stream)
    be2.stage_ms = None
    rows = int(max(plan["rows"]))
    eng2 = be2.engine
    att_ms = eng2.probe(2, min(rows, eng2.max_batch), 16)
    att_fl = 4.0 * min(rows, eng2.max_batch) * d2.n_audio_head * 1500 * 1500 * 64
    att_tf = att_fl / (att_ms * 1e-3) / 1e12
    enc_ms = stages.get("encode", 0.0)
    enc_tf = encoder_flops(d2) * 60 / (enc_ms * 1e-3) / 1e12 if enc_ms else None
    return {"value": round(1800.0 / dt, 2), "unit": "x realtime (audio s / wall s)", "model": "tiny (d 384, 6 heads, 4 + 4 layers, vocab 51865)",
            "batch_size": B2, "chunks": 60, "audio_s": 1800.0, "wall_ms": round(dt * 1e3, 2), "runs_ms": [round(t * 1e3, 2) for t in ts],
            "rows_per_pass": plan["rows"], "passes_in_flight": plan["passes_in_flight"], "forced_tokens": tokens,
            "stages_ms_summed_over_passes": stages,
            "roofline": {"kernel": "attn_full_kernel", "bound": "mfma", "achieved": round(att_tf, 1), "peak": MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(att_tf / MFMA_PEAK_TFLOPS, 4), "traffic": None,
                         "avg_launch_us": round(att_ms * 1e3, 2), "rows_per_launch": min(rows, eng2.max_batch),
                         "algorithmic_flops_per_launch": att_fl,
                         "duration_source": "live HIP-event probe on the engine's stream, launches back to back (one layer's attention of the widest pass)"},
            "encoder_mfma": {"achieved_TFLOPs": round(enc_tf, 1) if enc_tf else None,
                             "frac": round(enc_tf / MFMA_PEAK_TFLOPS, 4) if enc_tf else None,
                             "note": "37 GFLOP per chunk x 60 over the encoders' summed GPU time"},
            "workload": "whisper-tiny fp16 batch_size=8, 30 min synthetic 16 kHz audio as 60 chunks in one transcribe_batch call: "
                        "log-mel + encoder + greedy decode + cross-attention DTW + result dicts; input resident in HBM"}


def vad_mix(be, audio, wt, B, dev):
    """Config 4's shape of input: VAD chunks instead of fixed windows.  The 30-minute synthetic file is cut where the
    reference's own run cut its 30-minute file (81 windows of 1.2 .. 30 s, tests/golden/gold30m_windows.json) and every
    chunk is decoded to the token count the reference produced for that window (4 .. 199, mean 108): ragged chunks,
    rows that end at different steps (finished rows sit out, longest chunks are scheduled first).  Reported beside
    `value`; same API call."""
    segs, lens, secs = _vad_segments(torch.from_numpy(audio).to(dev))
    kw = dict(batch_size=B, language="en", word_timestamps=wt, forced_len=max(lens), forced_lens=lens, return_chunks=True)
    dt, res, ts = _best_of(lambda: be.transcribe_batch(segs, **kw), dev)
    assert [len(c["tokens"]) for c in res["chunks"]] == lens
    return {"value": round(secs / dt, 2), "unit": "x realtime (audio s / wall s)", "chunks": len(segs), "audio_s": round(secs, 1),
            "mean_chunk_s": round(secs / len(segs), 2), "mean_tokens": round(float(np.mean(lens)), 1), "wall_ms": round(dt * 1e3, 1),
            "runs_ms": [round(t * 1e3, 1) for t in ts],
            "workload": "81 VAD-shaped chunks cut at the reference run's window starts, per-chunk token counts of that run"}


ALIGN_LABELS = ["<pad>", "<s>", "</s>", "<unk>", "|"] + list("etaonihsrdlumwcfgypbvk'xjqz")      # wav2vec2-base-960h's vocabulary


def _bench_align_model(be, dev):
    """wav2vec2-base with seeded random weights and wav2vec2-base-960h's vocabulary, registered as the backend's "en" align model"""
    from whisperx_mlx_amd.w2v import W2VConfig, W2VHipModel, pack_w2v, random_state_dict
    wcfg = W2VConfig()
    m = W2VHipModel(wcfg, pack_w2v(random_state_dict(wcfg, seed=1), wcfg, dev), device_index=dev.index or 0)
    meta = {"language": "en", "dictionary": {c.lower(): i for i, c in enumerate(ALIGN_LABELS)}, "type": "hip"}
    be.align_model_cache["align_en"] = (m, meta)


def config4_sharded(be, audio, B, dev, n_gpus, rank, dist, dist_backend):
    """BASELINE.json config 4 AS WRITTEN, over the ranks of this run: the VAD-shaped chunks (81 per GPU: weak scaling)
    sharded over the GPUs, every rank transcribes (large-v3) and force-aligns (wav2vec2-base) its share, ONE gather of the
    fixed-width records carries tokens and aligned words, every rank rebuilds the aligned result dict
    (parallel.transcribe_batch_sharded).  Timed between barriers, maximum over the ranks."""
    _bench_align_model(be, dev)
    segs1, lens1, secs1 = _vad_segments(torch.from_numpy(audio).to(dev))
    segs, lens = [], []
    for r in range(n_gpus):                     # the same 81 windows once per GPU, on a common time axis
        segs += [dict(s_, start=s_["start"] + 1800.0 * r, end=s_["end"] + 1800.0 * r) for s_ in segs1]
        lens += lens1
    # materialize="lazy": every rank ends with the job's record table and a result whose dicts exist for its OWN chunks
    # (built by its launcher threads and its align() inside the timed region, like a single-GPU job's) and are built from
    # the records for the other ranks' chunks when they are read -- `read_all_ms` below, outside the timed region
    kw = dict(batch_size=B, align_words=True, language="en", forced_len=max(lens), forced_lens=lens, materialize="lazy")
    from whisperx_mlx_amd import parallel as PAR_
    PAR_.transcribe_batch_sharded(be, segs, **kw)             # graphs of the rank's launch shapes
    ts, tms = [], []
    for _ in range(3):
        torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        tm = {}
        res = PAR_.transcribe_batch_sharded(be, segs, timings=tm, **kw)
        torch.cuda.synchronize(dev)
        dist.barrier()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if dist_backend == "nccl" else "cpu")
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        ts.append(float(dt.item()))
        tms.append(tm)
    k = int(np.argsort(ts)[1])
    dt, tm = ts[k], tms[k]
    t0 = time.perf_counter()
    words = [w for s_ in res["segments"] for w in s_.get("words", [])]
    read_all = time.perf_counter() - t0
    return {"value": round(n_gpus * secs1 / dt, 2), "unit": "x realtime (audio s / wall s)", "n_gpus": n_gpus, "chunks": len(segs),
            "audio_s": round(n_gpus * secs1, 1), "wall_ms": round(dt * 1e3, 1), "runs_ms": [round(t * 1e3, 1) for t in ts],
            "aligned_segments": len(res["segments"]), "aligned_words": len(words),
            "rank0_ms": {"own_share": round(tm["local"] * 1e3, 1), "pack": round(tm["pack"] * 1e3, 2),
                         "gather_incl_waiting_for_the_slowest_rank": round(tm["gather"] * 1e3, 2), "table_and_result": round(tm["assemble"] * 1e3, 2)},
            "read_all_ms": round(read_all * 1e3, 1),
            "workload": f"{len(segs)} VAD-shaped chunks sharded over {n_gpus} ranks -> large-v3 -> wav2vec2-base forced alignment on the "
                        "rank that holds the chunk -> one gather of tokens + aligned words -> the record table + the aligned result "
                        "(own chunks as dicts, the others built from their records on access) on every rank"}


def config4(be, audio, B, dev):
    """BASELINE.json config 4 on one GPU, END TO END in one timed region: VAD-shaped chunks -> whisper-large-v3 ->
    wav2vec2-base forced alignment (CTC forward of every chunk's audio, trellis + beam-2 backtrack, char -> word ->
    sentence assembly) -> the reference's aligned result dict, through `transcribe_batch(..., align_words=True)`
    (whisperx/asr.py:50-87 -> backends/mlx_lightning.py:290-369 -> alignment.py:113-380).  Seeded random weights for both
    models; the transcript is what the random Whisper emits (its token ids spelled out: ~6 characters per token, a third
    more alignment targets than real text)."""
    _bench_align_model(be, dev)
    segs, lens, secs = _vad_segments(torch.from_numpy(audio).to(dev))
    kw = dict(batch_size=B, language="en", forced_len=max(lens), forced_lens=lens)
    dt_asr, _res, _ts = _best_of(lambda: be.transcribe_batch(segs, **kw), dev)
    dt, res, ts = _best_of(lambda: be.transcribe_batch(segs, align_words=True, **kw), dev)
    words = [w for s_ in res["segments"] for w in s_.get("words", [])]
    failed = sum(1 for s_ in res["segments"] if s_.get("chars", 0) is None and not s_["words"])
    return {"value": round(secs / dt, 2), "unit": "x realtime (audio s / wall s)", "chunks": len(segs), "audio_s": round(secs, 1),
            "wall_ms": round(dt * 1e3, 1), "runs_ms": [round(t * 1e3, 1) for t in ts],
            "asr_only_ms": round(dt_asr * 1e3, 1), "align_stage_ms": round((dt - dt_asr) * 1e3, 1),
            "aligned_segments": len(res["segments"]), "aligned_words": len(words),
            "words_with_times": sum(1 for w in words if "start" in w), "segments_align_failed": failed,
            "workload": "81 VAD-shaped chunks (1 797 s) -> large-v3 (per-chunk token counts of the reference run) -> wav2vec2-base "
                        "CTC forward + forced alignment -> aligned word dicts; one transcribe_batch(align_words=True) call"}


def w2v_flops(cfg, n_samples):
    """multiply-adds x 2 of one wav2vec2 CTC forward over n_samples of audio"""
    T, c_in, fl = n_samples, 1, 0.0
    for k, st in zip(cfg.conv_kernel, cfg.conv_stride):
        T = (T - k) // st + 1
        fl += 2.0 * T * cfg.conv_dim * c_in * k
        c_in = cfg.conv_dim
    d = cfg.hidden
    fl += 2.0 * T * cfg.conv_dim * d                                         # feature projection
    fl += 2.0 * T * d * (d // cfg.pos_groups) * cfg.pos_kernel                # grouped positional conv
    fl += cfg.layers * (2.0 * T * d * d * 4 + 4.0 * T * T * d + 4.0 * T * d * cfg.ffn)
    fl += 2.0 * T * d * cfg.vocab
    return fl


def align_stage(be, chunks_dev, B, dev):
    """config 4's second model: wav2vec2-base CTC forward + trellis / beam-2 backtrack for 64 chunks of 30 s -- the batch
    `alignment.align` forwards at a time -- (seeded random weights, 400 target characters per chunk), timed with HIP
    events on the aligner's stream and reported per 16 chunks."""
    B = 64
    from whisperx_mlx_amd.w2v import W2VConfig, W2VHipModel, pack_w2v, random_state_dict
    wcfg = W2VConfig()
    m = W2VHipModel(wcfg, pack_w2v(random_state_dict(wcfg, seed=1), wcfg, dev), device_index=dev.index or 0)
    g = torch.Generator().manual_seed(7)
    align_tok = torch.randint(1, wcfg.vocab, (B, 400), generator=g, dtype=torch.int32).to(dev)
    align_N = torch.full((B,), 400, dtype=torch.int32, device=dev)
    pcm = chunks_dev[torch.arange(B, device=dev) % chunks_dev.shape[0]].contiguous()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    out = {}
    for it in range(3):
        with torch.cuda.stream(m.stream):
            ev[0].record(m.stream)
            logp, T = m.emissions_device(pcm, [480000] * B)
            ev[1].record(m.stream)
            m.ctc_align(logp, torch.tensor(T, dtype=torch.int32), align_tok, align_N, 0, 2)
            ev[2].record(m.stream)
        torch.cuda.synchronize(dev)
        out = {"w2v_forward_ms": round(ev[0].elapsed_time(ev[1]) * 16 / B, 3), "ctc_align_ms": round(ev[1].elapsed_time(ev[2]) * 16 / B, 3),
               "per": "16 chunks of 30 s", "segments_per_forward": B}
    secs = 16 * 30.0
    flops = 1.4e10 * secs          # ~1.4e10 FLOP per aligned audio second
    out["chunks"] = B
    out["w2v_TFLOPs"] = round(flops / (out["w2v_forward_ms"] * 1e-3) / 1e12, 1)
    out["w2v_mfma_frac"] = round(out["w2v_TFLOPs"] / MFMA_PEAK_TFLOPS, 4)
    # the same with the forward's FLOPs counted layer by layer (SURVEY's per-second figure is ~15 % low): conv stack + feature
    # projection + grouped positional conv + 12 x (QKV/out projections, attention over T frames, FFN) + lm-head
    fl1 = w2v_flops(wcfg, 480000)
    out["w2v_TFLOPs_counted"] = round(16 * fl1 / (out["w2v_forward_ms"] * 1e-3) / 1e12, 1)
    out["w2v_mfma_frac_counted"] = round(out["w2v_TFLOPs_counted"] / MFMA_PEAK_TFLOPS, 4)
    out["GFLOP_per_30s_segment"] = round(fl1 / 1e9, 1)
    out["model"] = f"wav2vec2-base (random weights), {B} x 30 s per forward, 400 target characters per chunk"
    return out


def longform(args, be, dims, n_gpus, rank, use_dist, dist, dev):
    """config 5: HOURS of synthetic long-form audio per GPU through batch_processor.batch_transcribe (30 s chunks with
    0.5 s overlap -> passes of the hot path -> merge), the reference's whisperx/batch_processor.py:279-338 flow."""
    from whisperx_mlx_amd.synth import speechlike_audio
    from whisperx_mlx_amd.batch_processor import batch_transcribe
    secs = args.longform * 3600.0
    minute = speechlike_audio(600.0, seed=1234 + rank)
    audio = np.tile(minute, int(np.ceil(secs / 600.0)))[: int(secs * 16000)]
    opts = {"language": "en", "forced_len": args.tokens}
    warm = 30 * args.batch * be._default_lanes()
    batch_transcribe(audio[: 16000 * warm], [{"start": 0.0, "end": float(warm)}], be, batch_size=args.batch, decode_options=opts)   # graphs
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    out = batch_transcribe(audio, [{"start": 0.0, "end": secs}], be, batch_size=args.batch, decode_options=opts)
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert len(out) == 1 and out[0]["text"]
    n_chunks = int(np.ceil(secs / 29.5))
    return {"metric": f"real-time factor (x) {args.model} long-form ({args.longform:g} h per GPU, batch_processor path)",
            "value": round(n_gpus * secs / dt, 2), "unit": "x realtime (audio s / wall s)", "n_gpus": n_gpus, "steps": 1,
            "warmup": 1, "ms_per_step": round(dt * 1e3, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if args.compute_type == "float16" else "f16 (int8 decoder GEMV weights)",
            "data": f"synthetic 16 kHz long-form audio, seeded random weights, forced {args.tokens} sampled tokens per chunk",
            "config": {"workload": f"whisper-{args.model} {args.compute_type}, {args.longform:g} h synthetic long-form stream per GPU, "
                                   f"batch_processor.batch_transcribe (30 s chunks, 0.5 s overlap), batch_size={args.batch}",
                       "chunks": n_chunks, "passes_in_flight_per_gpu": be.passes_in_flight, "parallelism": f"dp{n_gpus}"}}


def real_run(args, be):
    """real checkpoint + the real 30-minute audio (north star acceptance clause): word-timestamp MAE of the decoder's
    cross-attention DTW words against the gold standard the reference ships (whisperx-large-v3-gold-standard/30m.json,
    committed as tests/golden/gold30m/30m.json.gz), the mean token count per 30 s, and the similarity of the greedy
    token ids with the reference's own large-v3 run (tests/golden/gold30m_windows.json).  Token parity against the
    oracle on real weights is tests/test_gpu_real_checkpoint.py (the oracle is a CPU restatement: minutes per window)."""
    import gzip
    from whisperx_mlx_amd import metrics as M
    from whisperx_mlx_amd.backend import load_audio
    audio = load_audio(args.audio)
    with gzip.open(os.path.join(ROOT, "tests", "golden", "gold30m", "30m.json.gz"), "rt") as f:
        gold = json.load(f)
    res = be.transcribe(audio, batch_size=args.batch, language=gold.get("language", "en"), word_timestamps="dtw",
                        return_chunks=True)
    m = M.word_mae_ms(M.flatten_words(res), M.flatten_words(gold))
    toks = [t for c in res.get("chunks", []) for t in c["tokens"]]
    with open(os.path.join(ROOT, "tests", "golden", "gold30m_windows.json")) as f:
        ref_toks = [t for w in json.load(f)["windows"] for t in w["tokens"]]
    ts0 = be.tokenizer.timestamp_begin          # windows are cut differently (fixed 30 s vs VAD): compare the text ids
    return {"word_mae_ms": m["mae_ms"], "word_mae": m, "mean_sampled_tokens_per_30s": round(len(toks) / (len(audio) / 480000.0), 1),
            "text_token_similarity_to_reference_run": round(M.token_similarity([t for t in toks if t < ts0], [t for t in ref_toks if t < ts0]), 4)}


def cpu_baseline(args, dims, packed, chunks, prompt, suppress):
    """The oracle (kind "port": our torch-CPU fp32 restatement) on the host cores at the workload's batch size:
    log-mel of 16 chunks, the encoder on ONE chunk (chunks are independent: x16 is exact), the cross-K/V projection and
    `cpu_tokens` real greedy steps at batch 16 (every later step costs the same: it is weight-bandwidth bound; the
    self-attention cache grows from 3 to 148 keys, negligible next to 6.4 GB of fp32 weights per step), scaled to the
    workload's token count."""
    from oracle import decoding as OD, logmel as OL, whisper_ref as OW
    from whisperx_mlx_amd import weights as W
    from whisperx_mlx_amd.audio import mel_filters
    share = host_cpu_share()
    if torch.get_num_threads() > share:
        torch.set_num_threads(share)               # the box's CPU share, not the host's CPU count (see host_cpu_share)
    threads = torch.get_num_threads()
    ck = W.random_checkpoint(dims, seed=0, std=0.02, device=packed["dec.emb"].device)
    w = {k: v.float().cpu() for k, v in ck.items()}
    del ck
    Bc = args.batch
    t0 = time.perf_counter()
    mel = OL.log_mel_chunks([chunks[i] for i in range(Bc)], [480000] * Bc, mel_filters(dims.n_mels))
    t1 = time.perf_counter()
    enc1 = OW.encoder_forward(w, dims, torch.from_numpy(mel[:1]))
    t2 = time.perf_counter()
    enc = enc1.repeat(Bc, 1, 1)
    sp = OD.Specials.for_vocab(dims.n_vocab)
    OD.greedy_decode(w, dims, enc, sp, prompt, rules=args.rules, suppress_tokens=suppress, forced_len=args.cpu_tokens)
    t3 = time.perf_counter()
    OW.cross_kv(w, dims, enc[:1])
    xkv = Bc * (time.perf_counter() - t3)                   # the decode above projected the cross K/V of all 16 rows once
    dec_run = t3 - t2
    steps_run = max(dec_run - xkv, 0.5 * dec_run)           # its `cpu_tokens` decoder calls (the first one takes the prompt)
    dec_full = (dec_run - steps_run) + steps_run * args.tokens / args.cpu_tokens
    wall = (t1 - t0) + Bc * (t2 - t1) + dec_full
    return {"value": round(Bc * 30.0 / wall, 3), "unit": "x realtime (audio s / wall s)", "cores": threads, "kind": "port",
            "host_cpus": os.cpu_count(), "cpu_share_of_this_box": share,
            "sample": f"batch {Bc}: log-mel of {Bc} chunks {t1 - t0:.2f}s + encoder of 1 chunk {t2 - t1:.2f}s (x{Bc}) + cross-K/V and "
                      f"{args.cpu_tokens} real greedy steps at batch {Bc} {dec_run:.2f}s, steps scaled to {args.tokens} tokens "
                      f"({dec_full:.2f}s); torch-CPU fp32, {threads} threads"}


if __name__ == "__main__":
    main()
