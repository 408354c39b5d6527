"""Engine contexts and the memory policy around them (the backend's scheduler keeps several passes in flight: one context
= one HIP stream + workspace + hipGraph cache + launcher thread, the packed weights shared).

A 128-row context of large-v3 holds ~38 GB of workspace (31.5 GB of it the cross K/V of its rows, 4.9 GB the self-attention
cache of 232 positions per sequence), a 16-row one ~4.8 GB; the encoder's activations (6.0 GB at 128 rows) exist once per
process and model, whatever the number of contexts (csrc/api.hip EncWs: handed from encoder to encoder through an event).
What this module decides:
  * the FIRST context is built for the rows the backend was asked for; when that does not fit (other processes, other
    models on the GPU) the rows are halved until it does (first_context);
  * contexts beyond the first are built when a job has enough passes to keep them busy, and sized by the job: jobs of
    <= 16-row passes get 16-row contexts; when a wide job comes, those are rebuilt at the first context's size and keep
    their streams (get_contexts);
  * a context that fits but leaves torch's allocator no room for a pass's tensors counts as not fitting (new_context);
  * a context that does not fit means fewer passes in flight from then on, not an error per call.
Host logic; tested on the CPU with a stand-in engine (tests/test_backend_host.py).
"""
import warnings
from typing import List

import torch

from .audio import N_SAMPLES
from .engine import WhisperHipEngine


def is_oom(e: BaseException) -> bool:
    return "memory" in str(e).lower()


def new_context(dims, packed, rows, device_index, heads):
    """an engine context, or a RuntimeError("... out of memory ...") when its workspace fits but leaves no room for the
    tensors a pass of `rows` rows needs beside it (PCM, log-mel, encoder output: ~8 MB a row, two passes enqueued) --
    a context that starves torch's allocator fails later, in the middle of a job"""
    eng = WhisperHipEngine(dims, packed, max_batch=rows, device_index=device_index, alignment_heads=heads)
    free, _total = torch.cuda.mem_get_info(eng.device)
    need = (1 << 30) + 2 * rows * (8 << 20)
    if free < need:
        eng.close()
        raise RuntimeError(f"out of memory: {free >> 20} MiB left beside a context of {rows} rows, {need >> 20} MiB wanted for its passes")
    return eng


def first_context(dims, packed, rows, device_index, heads):
    """the backend's first context: `rows` rows, or the largest halving of it (>= 16) whose workspace fits"""
    while True:
        try:
            return new_context(dims, packed, rows, device_index, heads)
        except RuntimeError as e:
            if not is_oom(e) or rows <= 16:
                raise
            warnings.warn(f"no memory for an engine context of {rows} rows ({e}); trying {max(16, rows // 2)}")
            rows = max(16, rows // 2)
            torch.cuda.empty_cache()


def get_contexts(be, n, rows=None):
    """the first `n` engine contexts of backend `be` (be.engines; be.engine is the first), created on demand.

    rows: the launch shape of the job that asks (None or > 16: full-size contexts).  A job of <= 16-row passes gets
    16-row contexts beyond the first (~4.8 GB each for large-v3 instead of ~38 GB at 128 rows); contexts built that way
    are rebuilt at the first context's size, on their old streams, when a wider job comes."""
    full = be.engine.max_batch
    want = full if (rows is None or rows > 16) else min(full, 16)

    def build(r):
        return new_context(be.dims, be.engine.packed, r, be.device_index, be.engine.alignment_heads)

    for k in range(1, min(n, len(be.engines))):
        old = be.engines[k]
        if old.max_batch >= want:
            continue
        stream = old.stream
        old.close()
        torch.cuda.empty_cache()
        try:
            new = build(full)
        except RuntimeError as e:
            if not is_oom(e):
                raise
            warnings.warn(f"no memory to rebuild engine context {k + 1} at {full} rows ({e}): {k} pass(es) in flight")
            for dead in be.engines[k + 1:]:
                dead.close()
            del be.engines[k:]
            be._no_more_contexts = True
            be.engine.side_by_side = min(getattr(be.engine, "side_by_side", 1), k)
            be.engine.side_by_side_tested = min(getattr(be.engine, "side_by_side_tested", 1), k)
            break
        new.stream = stream          # the stream _default_lanes found to run beside the others
        be.engines[k] = new
    while len(be.engines) < n and not getattr(be, "_no_more_contexts", False):
        try:
            be.engines.append(build(want))
        except RuntimeError as e:
            if not is_oom(e):
                raise
            warnings.warn(f"no memory for engine context {len(be.engines) + 1} of {want} rows ({e}): "
                          f"{len(be.engines)} pass(es) in flight")
            be._no_more_contexts = True      # fewer passes in flight from here on, not an error per call
    return be.engines[:n]


class PassSlot:
    """Pinned host buffers one pass of the hot path writes its results to (and stages host PCM from), plus the event
    that says they have landed.  Every engine context owns two: a launcher thread turns pass i into text while pass
    i + 1 runs."""

    def __init__(self, rows, dims, device):
        self.rows, self.dims = rows, dims
        self.event = torch.cuda.Event()
        self.n = self.n_prompt = self.n_sampled = 0
        self.lens: List[int] = []
        self.marks = None
        ld = dims.n_audio_ctx + dims.n_text_ctx // 2 + 4
        pin = lambda *shape, dtype=torch.int32: torch.zeros(*shape, dtype=dtype).pin_memory()   # noqa: E731
        self.flen_dev = torch.zeros(rows, dtype=torch.int32, device=device)      # per-row forced lengths (bench workload): a stable address for the hipGraph
        self._h = {"nv": pin(rows), "flen": pin(rows), "tokens": pin(rows, dims.n_text_ctx), "sum_lp": pin(rows, dtype=torch.float32),
                   "nsp": pin(rows, dtype=torch.float32), "n_rows": pin(rows), "pi": pin(rows, ld), "pj": pin(rows, ld),
                   "plen": pin(rows)}

    def host(self, n, want_pcm=False):
        assert n <= self.rows
        if want_pcm and "pcm" not in self._h:          # only callers that hand over host arrays pay for the staging buffer
            self._h["pcm"] = torch.zeros(self.rows, N_SAMPLES, dtype=torch.float32).pin_memory()
        return self._h
